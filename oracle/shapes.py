"""Oracle: block shapes as 2-D outlines (x, z) extruded along y.

Restates ``Shape.from_urdf / from_mesh`` (assembly_gym/assembly_gym/envs/
assembly_env.py:45-68), ``merge_coplanar_faces`` (assembly_gym/assembly_gym/
utils/geometry.py:9-21) and ``Shape.get_face_frame_2d`` (assembly_env.py:118-124).

Conventions (see DESIGN.md "Arithmetic contract"):
* a shape is a convex polygon in the (x, z) plane with ``depth`` along y;
* 2-D face ``f`` is the directed edge ``(ia, ib)``; its frame is
  point = (v[ia] + v[ib]) * 0.5, x-axis = d / |d| with d = v[ib] - v[ia],
  outward normal n = (-d.z, d.x) / |d|   (assembly_env.py:118-124:
  xaxis = -cross(normal, y) = (n.z, -n.x), i.e. the same thing);
* the face INDEX order is the reference's: coplanar STL triangles are merged
  popping from the END of the face list and merged faces get increasing keys
  (geometry.py:9-21), so the 2-D faces appear in descending order of the
  largest STL triangle index of their plane; box URDFs use compas' ``Box``
  face order filtered by |n.y| < 1e-6 (assembly_env.py:50):
  bottom, +x, -x, top.
"""
import math
import os
import struct
import xml.etree.ElementTree as ET

import numpy as np

# float32 values exactly as stored in the reference's binary STL files
# (assembly_gym/shapes/blocks/{trapezoid,hexagon}.stl), widened to float64.
TRAP_ZB = -0.3595713675022125
TRAP_ZT = 0.5064539909362793
HEX_H = 0.8660253882408142
HEX_Z0 = -2.974833642933041e-17


def _box(side_x, side_z, depth):
    hx, hz = side_x / 2.0, side_z / 2.0
    return dict(
        verts=[(-hx, -hz), (-hx, hz), (hx, hz), (hx, -hz)],
        # compas Box faces with |n.y|<1e-6, in Box face order: bottom, +x, -x, top
        faces=[(3, 0), (2, 3), (0, 1), (1, 2)],
        depth=depth,
    )


SHAPES = {
    # faces: 0 left incline, 1 top (short), 2 right incline, 3 bottom (long)
    "trapezoid": dict(
        verts=[(-1.0, TRAP_ZB), (-0.5, TRAP_ZT), (0.5, TRAP_ZT), (1.0, TRAP_ZB)],
        faces=[(0, 1), (1, 2), (2, 3), (3, 0)],
        depth=1.0,
    ),
    # faces: 0 bottom, 1 lower-left, 2 lower-right, 3 upper-left, 4 top, 5 upper-right
    "hexagon": dict(
        verts=[(0.5, -HEX_H), (-0.5, -HEX_H), (-1.0, HEX_Z0), (-0.5, HEX_H), (0.5, HEX_H), (1.0, HEX_Z0)],
        faces=[(0, 1), (1, 2), (5, 0), (2, 3), (3, 4), (4, 5)],
        depth=1.0,
    ),
    "cube": _box(1.0, 1.0, 1.0),      # shapes/cube.urdf  <box size="1.0 1.0 1.0"/>
    "cube1": _box(1.0, 1.0, 1.0),     # shapes/cube1.urdf <box size="1.0 1.0 1.0"/>
    "cube06": _box(0.6, 0.6, 0.6),    # shapes/cube06.urdf <box size="0.6 0.6 0.6"/>
    "block": _box(0.10, 0.05, 0.05),  # shapes/block.urdf <box size="0.10 0.05 0.05"/>
}


class ShapeDef:
    """Immutable shape record used by every oracle module."""

    def __init__(self, name, verts, faces, depth, receiving_faces_2d=None, target_faces_2d=None):
        self.name = name
        self.verts = [(float(x), float(z)) for x, z in verts]
        self.faces = [(int(a), int(b)) for a, b in faces]
        self.depth = float(depth)
        self._receiving = list(receiving_faces_2d) if receiving_faces_2d else None
        self._target = list(target_faces_2d) if target_faces_2d else None
        self.area = polygon_area(self.verts)
        self.centroid = polygon_centroid(self.verts)

    @property
    def num_faces_2d(self):
        return len(self.faces)

    @property
    def target_faces_2d(self):          # assembly_env.py:85-87
        return self._target or list(range(self.num_faces_2d))

    @property
    def receiving_faces_2d(self):       # assembly_env.py:89-91
        return self._receiving or list(range(self.num_faces_2d))

    def face_frame_local(self, f):
        """(point, xaxis, normal) of face ``f`` in shape-local coordinates."""
        ia, ib = self.faces[f]
        return edge_frame(self.verts[ia], self.verts[ib])


def edge_frame(va, vb):
    """Frame of the directed edge va->vb.  assembly_env.py:118-124.

    Arithmetic contract (mirrored op-for-op by the HIP kernels):
      c = (va + vb) * 0.5 ; d = vb - va ; L = sqrt(dx*dx + dz*dz)
      t = d / L ; n = (-t.z, t.x)
    """
    cx = (va[0] + vb[0]) * 0.5
    cz = (va[1] + vb[1]) * 0.5
    dx = vb[0] - va[0]
    dz = vb[1] - va[1]
    L = math.sqrt(dx * dx + dz * dz)
    tx = dx / L
    tz = dz / L
    return (cx, cz), (tx, tz), (-tz, tx)


def polygon_area(verts):
    """Unsigned shoelace area (vertices are listed clockwise)."""
    s = 0.0
    n = len(verts)
    for i in range(n):
        x0, z0 = verts[i]
        x1, z1 = verts[(i + 1) % n]
        s += x0 * z1 - x1 * z0
    return abs(s) * 0.5


def polygon_centroid(verts):
    """Area centroid of the outline (= volume centroid of the prism)."""
    a = 0.0
    cx = 0.0
    cz = 0.0
    n = len(verts)
    for i in range(n):
        x0, z0 = verts[i]
        x1, z1 = verts[(i + 1) % n]
        w = x0 * z1 - x1 * z0
        a += w
        cx += (x0 + x1) * w
        cz += (z0 + z1) * w
    return (cx / (3.0 * a), cz / (3.0 * a))


def get_shape(name, receiving_faces_2d=None, target_faces_2d=None):
    d = SHAPES[name]
    return ShapeDef(name, d["verts"], d["faces"], d["depth"], receiving_faces_2d, target_faces_2d)


def shape_from_urdf_name(urdf_file, **kw):
    """'shapes/trapezoid.urdf' -> table entry (assembly_env.py:54-68)."""
    stem = os.path.splitext(os.path.basename(urdf_file))[0]
    if stem not in SHAPES:
        raise FileNotFoundError(f"URDF file not found: {urdf_file}")
    return get_shape(stem, **kw)


# --------------------------------------------------------------------------
# Mesh route: derive the same table from a URDF + binary STL on disk.  Used by
# tests (when /root/reference is present) to prove the hard-coded tables above
# are what the reference's loader would produce.
# --------------------------------------------------------------------------

def read_binary_stl(path):
    data = open(path, "rb").read()
    (n,) = struct.unpack("<I", data[80:84])
    tris = []
    for i in range(n):
        rec = struct.unpack("<12fH", data[84 + 50 * i: 84 + 50 * (i + 1)])
        tris.append([tuple(float(v) for v in rec[3 + 3 * j: 6 + 3 * j]) for j in range(3)])
    return tris


def outline_from_triangles(tris, tol=1e-6):
    """Plane-group the triangles and order the groups as merge_coplanar_faces
    (geometry.py:9-21) would: descending largest triangle index (single,
    unmerged triangles would keep their original key and come first)."""
    planes = []  # (normal, offset, [tri idx])
    for i, tri in enumerate(tris):
        p0, p1, p2 = (np.array(p) for p in tri)
        n = np.cross(p1 - p0, p2 - p0)
        n = n / np.linalg.norm(n)
        off = float(np.dot(n, p0))
        for pl in planes:
            if np.dot(pl[0], n) > 1 - tol and abs(pl[1] - off) < tol:
                pl[2].append(i)
                break
        else:
            planes.append((n, off, [i]))
    singles = sorted([pl for pl in planes if len(pl[2]) == 1], key=lambda pl: pl[2][0])
    merged = sorted([pl for pl in planes if len(pl[2]) > 1], key=lambda pl: -max(pl[2]))
    faces2d = []
    for n, off, idx in singles + merged:
        if abs(n[1]) < 1e-6:                      # assembly_env.py:50
            pts = {(p[0], p[2]) for i in idx for p in tris[i]}
            assert len(pts) == 2, pts
            a, b = sorted(pts)
            d = (b[0] - a[0], b[1] - a[1])
            # orient so that (-d.z, d.x) is the outward normal (n.x, n.z)
            if (-d[1]) * n[0] + d[0] * n[2] < 0:
                a, b = b, a
            faces2d.append((a, b))
    depth = max(p[1] for t in tris for p in t) - min(p[1] for t in tris for p in t)
    # chain the directed edges into a clockwise vertex loop
    verts = []
    for a, b in faces2d:
        for p in (a, b):
            if p not in verts:
                verts.append(p)
    nxt = {a: b for a, b in faces2d}
    loop = [faces2d[0][0]]
    while nxt[loop[-1]] != loop[0]:
        loop.append(nxt[loop[-1]])
    faces = [(loop.index(a), loop.index(b)) for a, b in faces2d]
    return loop, faces, depth


def shape_from_urdf_file(urdf_path):
    """URDF -> (verts, faces, depth) by the mesh route."""
    root = ET.parse(urdf_path).getroot()
    geom = root.find("./link/collision/geometry")
    box = geom.find("box")
    if box is not None:
        sx, sy, sz = (float(v) for v in box.attrib["size"].split())
        d = _box(sx, sz, sy)
        return d["verts"], d["faces"], d["depth"]
    mesh = geom.find("mesh").attrib["filename"]
    assert mesh.startswith("package://")
    stl = os.path.join(os.path.dirname(urdf_path), mesh[len("package://"):])
    return outline_from_triangles(read_binary_stl(stl))
