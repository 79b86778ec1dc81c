"""Host side of the shape table: URDF/STL -> 2-D outline -> ``bridges_shape``.

Mirrors what ``Shape.from_urdf`` / ``Shape.from_mesh`` / ``merge_coplanar_faces``
produce in the reference (assembly_gym/assembly_gym/envs/assembly_env.py:45-68,
assembly_gym/assembly_gym/utils/geometry.py:9-21):

* a URDF names either a ``<box size="x y z"/>`` or a binary STL mesh
  (``package://blocks/<name>.stl`` relative to the URDF);
* the mesh's coplanar triangles are merged; merging pops triangles from the END
  of the face list and every merged face gets a fresh, larger key, so the faces
  come out ordered by the largest triangle index of their plane, descending;
* the 2-D faces are those with |n.y| < 1e-6 in that order; a box keeps compas'
  ``Box`` face order: bottom, +x, -x, top.

Only y-extruded convex prisms are supported (everything the path uses).
"""
import math
import os
import struct
import xml.etree.ElementTree as ET

from . import abi

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASSET_ROOT = os.path.join(PKG_ROOT, "assembly_gym")     # '<pkg>/assembly_gym/shapes/*.urdf'


class ShapeGeometry:
    """Outline + derived constants of one shape, ready for upload."""

    def __init__(self, verts, faces, depth, name=""):
        if not (3 <= len(verts) <= abi.MAX_VERTS) or len(faces) != len(verts):
            raise ValueError(f"shape '{name}': need a closed outline with 3..{abi.MAX_VERTS} faces")
        self.name = name
        self.verts = [(float(x), float(z)) for x, z in verts]
        self.faces = [(int(a), int(b)) for a, b in faces]
        self.depth = float(depth)
        # face frames in shape-local coordinates (arithmetic contract of DESIGN.md)
        self.face_centre, self.face_tangent, self.face_normal = [], [], []
        for a, b in self.faces:
            (ax, az), (bx, bz) = self.verts[a], self.verts[b]
            cx = (ax + bx) * 0.5
            cz = (az + bz) * 0.5
            dx = bx - ax
            dz = bz - az
            L = math.sqrt(dx * dx + dz * dz)
            tx = dx / L
            tz = dz / L
            self.face_centre.append((cx, cz))
            self.face_tangent.append((tx, tz))
            self.face_normal.append((-tz, tx))
        acc = gx = gz = 0.0
        n = len(self.verts)
        for i in range(n):
            x0, z0 = self.verts[i]
            x1, z1 = self.verts[(i + 1) % n]
            w = x0 * z1 - x1 * z0
            acc += w
            gx += (x0 + x1) * w
            gz += (z0 + z1) * w
        self.area = abs(acc) * 0.5
        self.centroid = (gx / (3.0 * acc), gz / (3.0 * acc))
        self.volume = self.area * self.depth

    @property
    def num_faces_2d(self):
        return len(self.faces)

    def to_struct(self):
        s = abi.Shape()
        s.nv = len(self.verts)
        for i, (x, z) in enumerate(self.verts):
            s.vx[i], s.vz[i] = x, z
        for f, (a, b) in enumerate(self.faces):
            s.fa[f], s.fb[f] = a, b
            s.fcx[f], s.fcz[f] = self.face_centre[f]
            s.fnx[f], s.fnz[f] = self.face_normal[f]
        s.depth, s.volume = self.depth, self.volume
        s.gx, s.gz = self.centroid
        return s


def _box_outline(sx, sy, sz):
    hx, hz = sx / 2.0, sz / 2.0
    verts = [(-hx, -hz), (-hx, hz), (hx, hz), (hx, -hz)]
    faces = [(3, 0), (2, 3), (0, 1), (1, 2)]          # bottom, +x, -x, top
    return verts, faces, sy


def _read_stl(path):
    with open(path, "rb") as fh:
        data = fh.read()
    (n,) = struct.unpack_from("<I", data, 80)
    if len(data) < 84 + 50 * n:
        raise ValueError(f"{path}: not a binary STL")
    tris = []
    for i in range(n):
        rec = struct.unpack_from("<12f", data, 84 + 50 * i)
        tris.append((rec[3:6], rec[6:9], rec[9:12]))
    return tris


def _prism_outline(tris, tol=1e-6):
    groups = []                                   # [unit normal, plane offset, [triangle indices]]
    for i, (p0, p1, p2) in enumerate(tris):
        u = [p1[k] - p0[k] for k in range(3)]
        v = [p2[k] - p0[k] for k in range(3)]
        n = [u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]]
        ln = math.sqrt(sum(c * c for c in n))
        n = [c / ln for c in n]
        off = sum(n[k] * p0[k] for k in range(3))
        for g in groups:
            if sum(g[0][k] * n[k] for k in range(3)) > 1 - tol and abs(g[1] - off) < tol:
                g[2].append(i)
                break
        else:
            groups.append([n, off, [i]])
    unmerged = sorted((g for g in groups if len(g[2]) == 1), key=lambda g: g[2][0])
    merged = sorted((g for g in groups if len(g[2]) > 1), key=lambda g: -max(g[2]))
    edges = []
    for n, _off, idx in unmerged + merged:
        if abs(n[1]) >= 1e-6:
            continue
        pts = sorted({(p[0], p[2]) for i in idx for p in tris[i]})
        if len(pts) != 2:
            raise ValueError("not a y-extruded prism")
        a, b = pts
        if (-(b[1] - a[1])) * n[0] + (b[0] - a[0]) * n[2] < 0:     # make (-d.z, d.x) the outward normal
            a, b = b, a
        edges.append((a, b))
    ys = [p[1] for t in tris for p in t]
    nxt = dict(edges)
    loop = [edges[0][0]]
    while nxt[loop[-1]] != loop[0]:
        loop.append(nxt[loop[-1]])
    faces = [(loop.index(a), loop.index(b)) for a, b in edges]
    return loop, faces, max(ys) - min(ys)


def resolve_urdf(urdf_file):
    """assembly_env.py:54-61: the path as given, else relative to the package."""
    if os.path.exists(urdf_file):
        return urdf_file
    cand = os.path.join(ASSET_ROOT, urdf_file)
    if os.path.exists(cand):
        return cand
    raise FileNotFoundError(f"URDF file not found: {urdf_file}")


_cache = {}


def load_urdf(urdf_file):
    path = os.path.abspath(resolve_urdf(urdf_file))
    if path in _cache:
        return _cache[path]
    geom = ET.parse(path).getroot().find("./link/collision/geometry")
    if geom is None:
        raise ValueError(f"{path}: no <collision><geometry>")
    box = geom.find("box")
    if box is not None:
        sx, sy, sz = (float(v) for v in box.attrib["size"].split())
        verts, faces, depth = _box_outline(sx, sy, sz)
    else:
        fn = geom.find("mesh").attrib["filename"]
        if fn.startswith("package://"):
            fn = os.path.join(os.path.dirname(path), fn[len("package://"):])
        verts, faces, depth = _prism_outline(_read_stl(fn))
    g = ShapeGeometry(verts, faces, depth, name=os.path.splitext(os.path.basename(path))[0])
    _cache[path] = g
    return g
