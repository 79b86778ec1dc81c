"""Oracle: 2-D rigid placement of blocks.

Restates ``align_frames_2d`` (assembly_gym/assembly_gym/utils/geometry.py:39-50),
``AssemblyGym.create_block`` (assembly_gym/assembly_gym/envs/gym_env.py:204-216),
``Block.__init__`` (assembly_gym/assembly_gym/envs/assembly_env.py:146-153),
``distance_box_point`` / ``project_point_on_box`` (geometry.py:89-105) and the
compas ``Box.contains_point(tol=1e-6)`` used by ``_update_targets``
(gym_env.py:163-169).

Arithmetic contract (every operation is a separately rounded IEEE-754 double
operation; no fused multiply-add; the HIP kernels follow the same order):

  rotation about +y by phi, stored as (c, s) = (cos phi, sin phi):
      R(v) = ( v.x*c + v.z*s ,  v.z*c - v.x*s )
  align (n1, t1, c1 = target frame in world; n2, c2 = new shape's face, local):
      dot  = n1.x*n2.x + n1.z*n2.z
      c    = clip(-dot, -1, 1)                          # cos(arccos(clip(-n1.n2)))
      cy   = n1.z*n2.x - n1.x*n2.z                       # (n1 x n2).y
      s    = |cy| if cy + 1e-6 >= 0 else -|cy|           # axis = n1 x n2 + (0,1e-6,0)
      r2   = R(c2)
      pos  = ((c1 + ox*t1) + oy*n1) - r2
      v_w  = pos + R(v_local)

  The reference goes angle = arccos(c) -> axis-angle matrix -> quaternion ->
  matrix (compas); (c, s) above are the exact cos/sin of that angle, so the
  two differ only by compas' rounding (unknowable offline).
"""
import math

from .shapes import edge_frame

FLOOR_FRAME = ((0.0, 0.0), (1.0, 0.0), (0.0, 1.0))   # point, xaxis, normal (assembly_env.py:339-340)


def rot(v, c, s):
    return (v[0] * c + v[1] * s, v[1] * c - v[0] * s)


def align(frame1, c2, n2, ox, oy):
    (c1x, c1z), (t1x, t1z), (n1x, n1z) = frame1
    dot = n1x * n2[0] + n1z * n2[1]
    c = -dot
    if c > 1.0:
        c = 1.0
    if c < -1.0:
        c = -1.0
    cy = n1z * n2[0] - n1x * n2[1]
    s = abs(cy) if cy + 1e-6 >= 0 else -abs(cy)
    r2x, r2z = rot(c2, c, s)
    px = ((c1x + ox * t1x) + oy * n1x) - r2x
    pz = ((c1z + ox * t1z) + oy * n1z) - r2z
    return (px, pz), (c, s)


class Block:
    """A posed shape (assembly_env.py:140-157)."""

    def __init__(self, shape, pos, cs=(1.0, 0.0)):
        self.shape = shape
        self.pos = (float(pos[0]), float(pos[1]))
        self.cs = (float(cs[0]), float(cs[1]))
        self.is_static = False
        c, s = self.cs
        self.verts = []
        for v in shape.verts:
            rx, rz = rot(v, c, s)
            self.verts.append((self.pos[0] + rx, self.pos[1] + rz))
        gx, gz = rot(shape.centroid, c, s)
        self.centroid = (self.pos[0] + gx, self.pos[1] + gz)
        self.frames = [edge_frame(self.verts[a], self.verts[b]) for a, b in shape.faces]
        xs = [v[0] for v in self.verts]
        zs = [v[1] for v in self.verts]
        self.aabb = (min(xs), min(zs), max(xs), max(zs))

    # assembly_env.py:118-124 on the transformed mesh
    def face_frame(self, f):
        return self.frames[f]

    @property
    def weight_per_density(self):
        return self.shape.area * self.shape.depth

    def aabb_contains(self, target, tol=1e-6):
        """compas Box.contains_point on mesh.aabb() (gym_env.py:166)."""
        x0, z0, x1, z1 = self.aabb
        tx, ty, tz = target
        cx = (x0 + x1) * 0.5
        cz = (z0 + z1) * 0.5
        hx = (x1 - x0) * 0.5
        hz = (z1 - z0) * 0.5
        hy = self.shape.depth * 0.5
        return (abs(tx - cx) < hx + tol) and (abs(ty) < hy + tol) and (abs(tz - cz) < hz + tol)

    def distance_to_point(self, target):
        """geometry.py:89-105."""
        if self.aabb_contains(target):
            return 0.0
        x0, z0, x1, z1 = self.aabb
        hy = self.shape.depth * 0.5
        qx = min(max(target[0], x0), x1)
        qy = min(max(target[1], -hy), hy)
        qz = min(max(target[2], z0), z1)
        return math.sqrt((target[0] - qx) ** 2 + (target[1] - qy) ** 2 + (target[2] - qz) ** 2)


def create_block(shapes, blocks, action):
    """gym_env.py:204-216.  ``action`` = (target_block, target_face, shape, face, ox, oy)."""
    tb, tf, sh, f, ox, oy = action[:6]
    frame1 = FLOOR_FRAME if tb == -1 else blocks[tb].face_frame(tf)
    shape = shapes[sh]
    c2, _t2, n2 = shape.face_frame_local(f)
    pos, cs = align(frame1, c2, n2, float(ox), float(oy))
    return Block(shape, pos, cs)
