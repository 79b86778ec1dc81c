"""Geometry helpers of the path (assembly_gym/assembly_gym/utils/geometry.py:39-50, 89-105 of the reference)."""
import math

from bridges_hip import ops


class Rotation2D(tuple):
    """Rotation about +y as (cos, sin), with the ``.quaternion`` the reference's callers read off compas' Rotation
    (``Block(shape, position, orientation=rotation.quaternion)``, notebooks/CRA_Assembly.ipynb cell 6)."""

    def __new__(cls, c, s):
        return super().__new__(cls, (float(c), float(s)))

    @property
    def quaternion(self):
        from assembly_gym.envs.assembly_env import Quaternion
        return Quaternion.from_cos_sin(self[0], self[1])      # keeps the kernel's own (cos, sin) for Block posing


def align_frames_2d(frame1, frame2, frame1_coordinates=None):
    """(position, rotation) aligning a shape face onto frame1 (geometry.py:39-50).  ``frame2`` is what
    ``shape.get_face_frame_2d(face)`` returned (a FaceFrame2D, which names its shape and face: the kernel takes the local
    face frame from the uploaded shape table) or the pair (shape, face) itself; ``rotation`` unpacks as (cos, sin) and has
    ``.quaternion``."""
    shape, face = (frame2.shape, frame2.face) if hasattr(frame2, "face") else frame2
    if frame1_coordinates is None:
        frame1_coordinates = [0, 0, 0]
    f1 = (frame1.point[0], frame1.point[2], frame1.xaxis[0], frame1.xaxis[2], frame1.normal[0], frame1.normal[2])
    pose, _verts = ops.place(f1, shape.geometry, face, frame1_coordinates[0], frame1_coordinates[2])
    return [pose[0], 0.0, pose[1]], Rotation2D(pose[2], pose[3])


def project_point_on_box(box, point):
    return (min(max(point[0], box.xmin), box.xmax),
            min(max(point[1], box.ymin), box.ymax),
            min(max(point[2], box.zmin), box.zmax))


def distance_box_point(box, point):
    if box.contains_point(point):
        return 0.
    q = project_point_on_box(box, point)
    return math.sqrt(sum((a - b) ** 2 for a, b in zip(point, q)))


def maximum_tension(forces):
    """geometry.py:132-143 of the reference: the largest net tension c_nn - c_np over all contact points (0 if none
    pulls).  ``forces``: [n_interfaces, points, >= 2] rows of (c_np, c_nn, ...) as ops.stability(..., tension_tol=)
    returns them (the reference walks the interfaces of a solved CRA assembly, which this build does not model)."""
    worst = 0.0
    for interface in forces:
        for point in interface:
            tension = float(point[0]) - float(point[1])
            if tension < 0:
                worst = max(worst, -tension)
    return worst
