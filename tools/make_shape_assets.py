"""Writes the shape assets shipped with the package
(bridges-with-reinforcement-learning_amd/assembly_gym/shapes/*.urdf, blocks/*.stl)
from numeric outline tables.

The outlines are the float32 coordinates of the reference's block meshes
(assembly_gym/shapes/blocks/{trapezoid,hexagon}.stl) and the <box> sizes of its
URDFs (assembly_gym/shapes/{cube,cube1,cube06,block}.urdf); the files themselves
are generated here, nothing is copied.  Triangles are emitted so that the
reference's face-merging order (geometry.py:9-21) yields the documented 2-D
face indices: -y cap, faces n-1 .. 0, +y cap.
"""
import os
import struct

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                    "bridges-with-reinforcement-learning_amd", "assembly_gym", "shapes")

ZB, ZT = np.float32(-0.3595713675022125), np.float32(0.5064539909362793)
H, Z0 = np.float32(0.8660253882408142), np.float32(-2.974833642933041e-17)
MESHES = {
    # 2-D faces as directed edges (a -> b), outward normal (-d.z, d.x), in face-index order
    "trapezoid": [((-1, ZB), (-.5, ZT)), ((-.5, ZT), (.5, ZT)), ((.5, ZT), (1, ZB)), ((1, ZB), (-1, ZB))],
    "hexagon": [((.5, -H), (-.5, -H)), ((-.5, -H), (-1, Z0)), ((1, Z0), (.5, -H)),
                ((-1, Z0), (-.5, H)), ((-.5, H), (.5, H)), ((.5, H), (1, Z0))],
}
BOXES = {"cube": "1.0 1.0 1.0", "cube1": "1.0 1.0 1.0", "cube06": "0.6 0.6 0.6", "block": "0.10 0.05 0.05"}

URDF = """<?xml version="1.0"?>
<robot name="{name}">
  <link name="{name}_base_link">
    <collision>
      <origin rpy="0 0 0" xyz="0 0 0"/>
      <geometry>
        {geom}
      </geometry>
    </collision>
  </link>
</robot>
"""


def prism_triangles(edges, hy=0.5):
    nxt = dict(edges)
    loop = [edges[0][0]]
    while nxt[loop[-1]] != loop[0]:
        loop.append(nxt[loop[-1]])          # clockwise seen from -y ... orientation handled by normals below
    tris = []

    def cap(y, flip):
        p0 = loop[0]
        for i in range(1, len(loop) - 1):
            a, b = loop[i], loop[i + 1]
            t = [(p0[0], y, p0[1]), (a[0], y, a[1]), (b[0], y, b[1])]
            tris.append(t[::-1] if flip else t)

    cap(-hy, flip=_cap_needs_flip(loop, -1))
    for a, b in reversed(edges):
        q = [(a[0], -hy, a[1]), (b[0], -hy, b[1]), (b[0], hy, b[1]), (a[0], hy, a[1])]
        n_out = (-(b[1] - a[1]), 0.0, b[0] - a[0])
        t1, t2 = [q[0], q[1], q[2]], [q[0], q[2], q[3]]
        if np.dot(np.cross(np.subtract(t1[1], t1[0]), np.subtract(t1[2], t1[0])), n_out) < 0:
            t1, t2 = t1[::-1], t2[::-1]
        tris += [t1, t2]
    cap(hy, flip=_cap_needs_flip(loop, +1))
    return tris


def _cap_needs_flip(loop, sign):
    p0, a, b = loop[0], loop[1], loop[2]
    n = np.cross((a[0] - p0[0], 0, a[1] - p0[1]), (b[0] - p0[0], 0, b[1] - p0[1]))
    return n[1] * sign < 0


def write_stl(path, tris):
    with open(path, "wb") as fh:
        fh.write(b"bridges-amd generated prism".ljust(80, b" "))
        fh.write(struct.pack("<I", len(tris)))
        for t in tris:
            n = np.cross(np.subtract(t[1], t[0]), np.subtract(t[2], t[0]))
            n = n / np.linalg.norm(n)
            fh.write(struct.pack("<12fH", *n, *t[0], *t[1], *t[2], 0))


if __name__ == "__main__":
    os.makedirs(os.path.join(ROOT, "blocks"), exist_ok=True)
    for name, edges in MESHES.items():
        edges = [((float(a[0]), float(a[1])), (float(b[0]), float(b[1]))) for a, b in edges]
        write_stl(os.path.join(ROOT, "blocks", name + ".stl"), prism_triangles(edges))
        open(os.path.join(ROOT, name + ".urdf"), "w").write(
            URDF.format(name=name, geom=f'<mesh filename="package://blocks/{name}.stl"/>'))
    for name, size in BOXES.items():
        open(os.path.join(ROOT, name + ".urdf"), "w").write(URDF.format(name=name, geom=f'<box size="{size}"/>'))
    print("wrote", sorted(os.listdir(ROOT)))
