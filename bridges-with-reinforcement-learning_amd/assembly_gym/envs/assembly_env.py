"""Shape / Block / AssemblyEnv with the reference's public surface
(assembly_gym/assembly_gym/envs/assembly_env.py:22-438 of the reference), restricted to the 2-D path the hot path
uses and backed by the HIP operators: posing and face frames (bridges_pose_block / bridges_face_frames),
stability (bridges_stability).  pybullet is not part of the path (assembly_env.py:164: pybullet_env=False by default,
successor_dqn.py:695 builds AssemblyEnv(render=False)); asking for it raises NotImplementedError."""
import math
from collections import namedtuple

import numpy as np

from bridges_hip import ops
from bridges_hip.shapes import ShapeGeometry, load_urdf

Frame2D = namedtuple("Frame2D", "point xaxis yaxis normal")     # 3-D tuples, y-axis = (0,1,0)


class FaceFrame2D(Frame2D):
    """Frame of a shape / block face as ``get_face_frame_2d`` returns it.  It remembers (shape, face): the placement
    kernel behind ``align_frames_2d`` takes the new block's face from the uploaded shape table, not from a frame."""

    def __new__(cls, shape, face, point, xaxis, yaxis, normal):
        self = super().__new__(cls, point, xaxis, yaxis, normal)
        self.shape, self.face = shape, face
        return self

    def to_world_coordinates(self, p):
        """compas Frame.to_world_coordinates for a local point (x along the face, y along +y, z along the normal)."""
        return [self.point[k] + p[0] * self.xaxis[k] + p[1] * self.yaxis[k] + p[2] * self.normal[k] for k in range(3)]


class Quaternion:
    """Minimal stand-in for compas.geometry.Quaternion (w, x, y, z); only rotations about y occur on the path."""

    def __init__(self, w=1.0, x=0.0, y=0.0, z=0.0):
        self.w, self.x, self.y, self.z = float(w), float(x), float(y), float(z)

    @classmethod
    def from_cos_sin(cls, c, s):
        phi = math.atan2(s, c)
        q = cls(math.cos(phi / 2.0), 0.0, math.sin(phi / 2.0), 0.0)
        q._cos_sin = (float(c), float(s))      # the (cos, sin) it was made from: a Block posed with it is bit for bit that pose
        return q

    def cos_sin(self):
        # rotation about +y by phi: x' = x cos + z sin  <=>  c = 1 - 2y^2, s = 2wy
        exact = getattr(self, "_cos_sin", None)
        return exact if exact is not None else (1.0 - 2.0 * self.y * self.y, 2.0 * self.w * self.y)

    @property
    def wxyz(self):
        return [self.w, self.x, self.y, self.z]

    def __iter__(self):
        return iter(self.wxyz)

    def __repr__(self):
        return f"Quaternion({self.w}, {self.x}, {self.y}, {self.z})"


class AABB:
    """mesh.aabb() as used by the path: contains_point (compas Box, tol 1e-6) and the extents."""

    def __init__(self, xmin, ymin, zmin, xmax, ymax, zmax):
        self.xmin, self.ymin, self.zmin, self.xmax, self.ymax, self.zmax = xmin, ymin, zmin, xmax, ymax, zmax

    def contains_point(self, p, tol=1e-6):
        cx, cy, cz = (self.xmin + self.xmax) * 0.5, (self.ymin + self.ymax) * 0.5, (self.zmin + self.zmax) * 0.5
        hx, hy, hz = (self.xmax - self.xmin) * 0.5, (self.ymax - self.ymin) * 0.5, (self.zmax - self.zmin) * 0.5
        return abs(p[0] - cx) < hx + tol and abs(p[1] - cy) < hy + tol and abs(p[2] - cz) < hz + tol


class Shape:
    """assembly_env.py:22-137.  ``urdf_file`` is resolved as given or relative to the package (shapes/*.urdf)."""

    def __init__(self, mesh=None, urdf_file=None, name="", receiving_faces_2d=None, target_faces_2d=None):
        self.urdf_file = None
        self.name = name
        self.geometry = None
        if mesh is not None:
            if not isinstance(mesh, ShapeGeometry):
                raise TypeError("mesh must be a bridges_hip.shapes.ShapeGeometry (compas meshes are not supported)")
            self.geometry = mesh
        elif urdf_file is not None:
            self.from_urdf(urdf_file)
        self._target_faces_2d = target_faces_2d
        self._receiving_faces_2d = receiving_faces_2d

    def from_urdf(self, urdf_file, package="blocks", merge_faces=True):
        from bridges_hip.shapes import resolve_urdf
        self.urdf_file = resolve_urdf(urdf_file)          # raises FileNotFoundError (assembly_env.py:59)
        self.geometry = load_urdf(self.urdf_file)

    @property
    def num_faces_2d(self):
        return self.geometry.num_faces_2d

    @property
    def faces_2d(self):
        return range(self.num_faces_2d)

    @property
    def target_faces_2d(self):
        return self._target_faces_2d or self.faces_2d

    @property
    def receiving_faces_2d(self):
        return self._receiving_faces_2d or self.faces_2d

    @property
    def verts_2d(self):
        return np.asarray(self.geometry.verts, dtype=np.float64)

    @property
    def vertices_2d(self):
        for x, z in self.verts_2d:
            yield [float(x), float(z)]

    @property
    def vertices(self):
        hy = self.geometry.depth / 2.0
        for y in (-hy, hy):
            for x, z in self.verts_2d:
                yield [float(x), y, float(z)]

    def _frames(self):
        g = self.geometry
        return [(g.face_centre[f], g.face_tangent[f], g.face_normal[f]) for f in range(g.num_faces_2d)]

    def get_face_frame_2d(self, face):
        c, t, n = self._frames()[face]
        return FaceFrame2D(self, face, point=(c[0], 0.0, c[1]), xaxis=(t[0], 0.0, t[1]), yaxis=(0.0, 1.0, 0.0),
                           normal=(n[0], 0.0, n[1]))

    def contains_2d(self, points):
        """assembly_env.py:126-137 on the HIP point-in-outline kernel."""
        return ops.contains_points(self, points)


class Block(Shape):
    """assembly_env.py:140-157: a shape with a position (x, y, z) and an orientation."""

    def __init__(self, shape, position, orientation=None, object_id=None, _posed=None):
        self.shape = shape
        self.geometry = shape.geometry
        self.name = shape.name
        self.urdf_file = shape.urdf_file
        self._target_faces_2d = None           # Block drops the shape's face restrictions (assembly_env.py:153)
        self._receiving_faces_2d = None
        self.position = [float(position[0]), float(position[1]), float(position[2])]
        self.object_id = object_id
        self.is_static = False
        self._bounding_box = None
        if _posed is not None:                 # (pose, verts, frames) straight from bridges_create_block
            self.pose, self._verts, self._frames_w = _posed
            self._orientation = orientation    # None: made from the pose's (cos, sin) when somebody asks (most candidates never do)
        else:
            self._orientation = orientation if orientation is not None else Quaternion(1., 0., 0., 0.)
            c, s = self._orientation.cos_sin() if hasattr(self._orientation, "cos_sin") else Quaternion(*self._orientation).cos_sin()
            self.pose = np.array([self.position[0], self.position[2], c, s], dtype=np.float64)
            self._verts, self._frames_w = ops.pose_block(self.geometry, self.pose)

    @property
    def orientation(self):
        if self._orientation is None:
            self._orientation = Quaternion.from_cos_sin(float(self.pose[2]), float(self.pose[3]))
        return self._orientation

    @property
    def bounding_box(self):
        """mesh.aabb() of the posed block -- computed when first asked for: a state's ~60 candidate blocks never are."""
        if self._bounding_box is None:
            hy = self.geometry.depth / 2.0
            xs, zs = self._verts[:, 0], self._verts[:, 1]
            self._bounding_box = AABB(xs.min(), self.position[1] - hy, zs.min(), xs.max(), self.position[1] + hy, zs.max())
        return self._bounding_box

    @property
    def verts_2d(self):
        return self._verts

    def _frames(self):
        return [((f[0], f[1]), (f[2], f[3]), (f[4], f[5])) for f in self._frames_w]

    def __repr__(self):
        return f"Block ({self.object_id})"


def is_stable_rbe(assembly_env):
    from assembly_gym.utils.stability import is_stable_rbe as f
    return f(assembly_env)


def _is_stable_rbe_variants(assembly_env, fixed_sets):
    from assembly_gym.utils.stability import is_stable_rbe as f
    return f.variants(assembly_env, fixed_sets)


is_stable_rbe.variants = _is_stable_rbe_variants


class _StateInfo(dict):
    """AssemblyEnv.state_info with the stability verdict computed on first access ('stable' / 'stability_info'), for the
    (blocks, frozen set) the environment held when the dictionary was made."""
    _LAZY = ("stable", "stability_info")

    def __init__(self, env, base):
        super().__init__(base)
        self._env = env
        self._snapshot = [(b, bool(b.is_static)) for b in env.blocks]
        self._solved = False

    def _solve(self, memo=True):
        if not self._solved:
            self._solved = True
            is_stable, info = self._env._solve_state(self._snapshot, memo=memo)
            dict.__setitem__(self, "stable", is_stable)
            dict.__setitem__(self, "stability_info", info)

    def __getitem__(self, key):
        if key in self._LAZY:
            self._solve()
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key in self._LAZY:
            self._solve()
        return dict.get(self, key, default)

    def __contains__(self, key):
        return key in self._LAZY or dict.__contains__(self, key)

    def _all(self):
        self._solve()
        return self

    def keys(self):
        return dict.keys(self._all())

    def items(self):
        return dict.items(self._all())

    def values(self):
        return dict.values(self._all())

    def __iter__(self):
        return dict.__iter__(self._all())

    def __repr__(self):
        return dict.__repr__(self._all())


class AssemblyEnv:
    """assembly_env.py:160-438 without the optional pybullet client."""

    def __init__(self, render=False, bounds=None, stability="rbe", mu=0.8, density=1.0, cra_env=True, pybullet_env=False):
        if pybullet_env or render:
            raise NotImplementedError("the pybullet physics client is outside the HIP hot path (SURVEY.md §2 #8)")
        if bounds is None:
            bounds = np.array([[-3., -3., -1], [7., 7., 9.]])
        self.bounds = np.asarray(bounds, dtype=np.float64)
        self.mu, self.density = mu, density
        self.client = None
        if stability == "rbe":
            self.stability_fct = is_stable_rbe
        elif stability == "rbe_penalty":                  # assembly_env.py:180-181 of the reference
            from assembly_gym.utils.stability import is_stable_rbe_penalty
            self.stability_fct = is_stable_rbe_penalty
        elif stability is None:
            self.stability_fct = lambda x: (None, None)
        elif callable(stability):
            self.stability_fct = stability
        else:
            raise NotImplementedError(f"stability='{stability}' is not built (only 'rbe' and 'rbe_penalty'; see assembly_gym/utils/stability.py)")
        self.cra_env = cra_env
        self.reset()

    def reset(self):
        self.obstacles, self.blocks = [], []
        self._stability_memo = {}
        self.is_block_frozen = False
        self.frozen_block_index = None
        self._update_state_info()

    @property
    def floor_half_width(self):
        return (self.bounds[1][0] - self.bounds[0][0]) / 2.0          # assembly_env.py:290-296

    @property
    def floor_depth(self):
        return self.bounds[1][1] - self.bounds[0][1]

    def _update_state_info(self):
        """assembly_env.py:306-331: the state dictionary incl. the stability verdict.  The verdict of a state that nobody
        reads is never computed (``add_block`` updates the state and ``AssemblyGym.step`` updates it again before anything
        looks: one of the reference's five solves per env-step), and a (blocks, frozen set) pair that was solved before
        is looked up -- the solver is a pure function of the pair (three of the remaining four)."""
        self._state_info = _StateInfo(self, {
            "last_block": self.blocks[-1] if self.blocks else None,
            "collision": False,                                        # no pybullet client (assembly_env.py:311-312)
            "collision_info": {"obstacles": [], "blocks": [], "floor": False, "bounding_box": False},
            "frozen_block": self.frozen_block_index,
        })

    def _solve_state(self, snapshot, memo=True):
        """(is_stable, info) of the assembly as it was when ``snapshot`` = [(block, is_static), ...] was taken."""
        blocks_now, flags_now = self.blocks, [b.is_static for b in self.blocks]
        same = len(snapshot) == len(blocks_now) and all(b is c and f == g for (b, f), c, g in zip(snapshot, blocks_now, flags_now))
        if not same:                                                   # evaluate the recorded state, then put the present one back
            self.blocks = [b for b, _ in snapshot]
            for b, f in snapshot:
                b.is_static = f
        try:
            key = (tuple(id(b) for b in self.blocks), tuple(bool(b.is_static) for b in self.blocks), float(self.mu),
                   float(self.density), id(self.stability_fct))
            cache = self._stability_memo
            if not memo or key not in cache:
                if len(cache) > 64:
                    cache.clear()
                variants = getattr(self.stability_fct, "variants", None)
                flags = key[1]
                if memo and variants is not None and self.blocks:
                    # the same blocks with the last one's frozen flag the other way round ride in the same operator call:
                    # AssemblyGym.step freezes the new block, solves, and stabilities_freezing() then asks for it free
                    # (gym_env.py:238-243, :325-333 of the reference) -- one launch and one copy back instead of two
                    other = flags[:-1] + (not flags[-1],)
                    sets = [{i for i, f in enumerate(fl) if f} for fl in (flags, other)]
                    first, second = variants(self, sets)
                    cache[key] = (first, list(self.blocks))
                    cache.setdefault((key[0], other) + key[2:], (second, list(self.blocks)))
                else:
                    cache[key] = (self.stability_fct(self), list(self.blocks))      # the blocks are kept alive: ids stay unique
            return cache[key][0]
        finally:
            if not same:
                self.blocks = blocks_now
                for b, f in zip(blocks_now, flags_now):
                    b.is_static = f

    def add_block(self, block):
        self.blocks.append(block)
        self._update_state_info()
        return self._state_info

    @property
    def state_info(self):
        return self._state_info

    def get_floor_frame(self):
        return Frame2D(point=(0.0, 0.0, 0.0), xaxis=(1.0, 0.0, 0.0), yaxis=(0.0, 1.0, 0.0), normal=(0.0, 0.0, 1.0))

    def add_obstacle(self, obstacle):
        self.obstacles.append(obstacle)

    def is_stable(self):
        return self._state_info["stable"]

    def freeze_block(self, block_index, freeze_color=None):
        self.blocks[block_index].is_static = True

    def unfreeze_block(self, block_index, default_color=None):
        self.blocks[block_index].is_static = False

    def disconnect_client(self):
        pass
