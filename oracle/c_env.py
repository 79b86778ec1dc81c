"""ctypes wrapper of oracle/c/liboracle_env.so -- the plain-C restatement of the hot path.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests to cross-check the numpy oracle with an
independent implementation, and by bench.py's ``cpu_baseline`` leg as the timed CPU port."""
import ctypes as C
import os
import subprocess

import numpy as np

from . import raster as R
from .env import OracleGym

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "c", "liboracle_env.so")
MAXV, MAXK, MAXG, MAXT, IMG = 6, 16, 24, 8, 64


class Shape(C.Structure):
    _fields_ = [("nv", C.c_int32), ("pad_", C.c_int32), ("vx", C.c_double * MAXV), ("vz", C.c_double * MAXV),
                ("fa", C.c_int32 * MAXV), ("fb", C.c_int32 * MAXV), ("fcx", C.c_double * MAXV), ("fcz", C.c_double * MAXV),
                ("fnx", C.c_double * MAXV), ("fnz", C.c_double * MAXV), ("depth", C.c_double), ("volume", C.c_double),
                ("gx", C.c_double), ("gz", C.c_double)]


class Cfg(C.Structure):
    _fields_ = [("max_steps", C.c_int32), ("a_max", C.c_int32), ("n_shapes", C.c_int32), ("n_groups", C.c_int32),
                ("group_shape", C.c_int32 * MAXG), ("group_face", C.c_int32 * MAXG),
                ("n_ground", C.c_int32), ("n_offsets", C.c_int32), ("n_targets", C.c_int32), ("pad_", C.c_int32),
                ("mu", C.c_double), ("density", C.c_double), ("floor_half_width", C.c_double), ("floor_depth", C.c_double),
                ("xlim", C.c_double * 2), ("ylim", C.c_double * 2), ("targets", (C.c_double * 3) * MAXT),
                ("x_ground", C.c_double * 32), ("offsets", C.c_double * 8), ("grid_x", C.c_double * IMG), ("grid_y", C.c_double * IMG),
                ("obstacle_bits", C.c_uint64 * IMG), ("reward_map", C.c_float * (IMG * IMG)), ("shapes", Shape * 8)]


class Cand(C.Structure):
    _fields_ = [("tb", C.c_int32), ("tf", C.c_int32), ("sh", C.c_int32), ("fc", C.c_int32), ("ox", C.c_double),
                ("pose", C.c_double * 4), ("verts", (C.c_double * 2) * MAXV), ("bits", C.c_uint64 * IMG), ("lin", C.c_float),
                ("inb", C.c_uint8), ("mask", C.c_uint8)]


class Out(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("valid_step", "action_index", "stable_frozen", "stable_unfrozen", "terminated",
                                         "truncated", "done", "no_actions", "n_blocks", "n_reached", "lp_pivots", "pad_")] + \
               [("reward", C.c_double), ("lin_reward", C.c_double), ("pose", C.c_double * 4)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(LIB)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(Cfg)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_reset.argtypes = [C.c_void_p]
        L.orc_candidates.restype = C.POINTER(Cand)
        L.orc_candidates.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_state_bits.restype = C.POINTER(C.c_uint64)
        L.orc_state_bits.argtypes = [C.c_void_p]
        L.orc_lockstep.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.POINTER(Out)]
        L.orc_run.restype = C.c_long
        L.orc_run.argtypes = [C.c_void_p, C.c_uint64, C.c_int32, C.c_long]
        L.orc_enable_f32.argtypes = [C.c_void_p]
        L.orc_f32.restype = C.POINTER(C.c_float)
        L.orc_f32.argtypes = [C.c_void_p]
        L.orc_blocks.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_double)]
        L.orc_candidate_stability.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
        L.orc_total_pivots.restype = C.c_long
        L.orc_total_pivots.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def bits_from_bool(img):
    return [int(sum(1 << q for q in np.flatnonzero(row))) for row in img]


def make_cfg(gym: OracleGym):
    """Configuration of the C env from a numpy OracleGym (task features are computed by the numpy oracle)."""
    cfg = Cfg()
    cfg.max_steps = gym.max_steps or 0
    groups = [(si, f) for si, s in enumerate(gym.shapes) for f in s.target_faces_2d]
    K = gym.max_steps or MAXK
    cfg.a_max = len(groups) * (len(gym.x_discr_ground) + K * MAXV * len(gym.offset_values))
    cfg.n_shapes, cfg.n_groups = len(gym.shapes), len(groups)
    for i, (si, f) in enumerate(groups):
        cfg.group_shape[i], cfg.group_face[i] = si, f
    cfg.n_ground, cfg.n_offsets, cfg.n_targets = len(gym.x_discr_ground), len(gym.offset_values), len(gym.targets)
    cfg.mu, cfg.density = gym.mu, gym.density
    cfg.floor_half_width = (gym.bounds[1][0] - gym.bounds[0][0]) / 2.0
    cfg.floor_depth = gym.bounds[1][1] - gym.bounds[0][1]
    cfg.xlim[0], cfg.xlim[1], cfg.ylim[0], cfg.ylim[1] = *gym.xlim, *gym.ylim
    for i, t in enumerate(gym.targets):
        for k in range(3):
            cfg.targets[i][k] = t[k]
    for i, v in enumerate(gym.x_discr_ground):
        cfg.x_ground[i] = v
    for i, v in enumerate(gym.offset_values):
        cfg.offsets[i] = v
    X, Y = R.pixel_grid(gym.xlim, gym.ylim, gym.img_size)
    for i in range(IMG):
        cfg.grid_x[i], cfg.grid_y[i] = X[i], Y[i]
    for r, b in enumerate(bits_from_bool(gym.obstacle_raster)):
        cfg.obstacle_bits[r] = b
    for i, v in enumerate(gym.reward_map.reshape(-1)):
        cfg.reward_map[i] = v
    for i, s in enumerate(gym.shapes):
        sh = cfg.shapes[i]
        sh.nv = len(s.verts)
        for k, (x, z) in enumerate(s.verts):
            sh.vx[k], sh.vz[k] = x, z
        for f, (a, b) in enumerate(s.faces):
            sh.fa[f], sh.fb[f] = a, b
            c, _t, n = s.face_frame_local(f)
            sh.fcx[f], sh.fcz[f], sh.fnx[f], sh.fnz[f] = c[0], c[1], n[0], n[1]
        sh.depth, sh.volume = s.depth, s.area * s.depth
        sh.gx, sh.gz = s.centroid
    return cfg


class CEnv:
    def __init__(self, gym: OracleGym):
        self.L = lib()
        self.cfg = make_cfg(gym)
        self.h = self.L.orc_create(C.byref(self.cfg))

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def candidates(self):
        n, nv = C.c_int32(), C.c_int32()
        p = self.L.orc_candidates(self.h, C.byref(n), C.byref(nv))
        return [p[i] for i in range(n.value)], nv.value

    def state_bits(self):
        p = self.L.orc_state_bits(self.h)
        return [p[i] for i in range(IMG)]

    def lockstep(self, seed, env_id):
        o = Out()
        self.L.orc_lockstep(self.h, seed, env_id, C.byref(o))
        return o

    def blocks(self):
        sh = (C.c_int32 * MAXK)()
        po = (C.c_double * (4 * MAXK))()
        n = self.L.orc_blocks(self.h, sh, po)
        return [(sh[b], (po[4 * b], po[4 * b + 1]), (po[4 * b + 2], po[4 * b + 3])) for b in range(n)]

    def candidate_stability(self):
        """is_action_stable_rbe (stability.py:122-130) of every candidate of the current state: uint8 [n_cand],
        0 for the masked-out ones."""
        n, nv = C.c_int32(), C.c_int32()
        self.L.orc_candidates(self.h, C.byref(n), C.byref(nv))
        out = (C.c_uint8 * max(n.value, 1))()
        self.L.orc_candidate_stability(self.h, out)
        return np.frombuffer(out, dtype=np.uint8, count=n.value).copy()

    def enable_f32(self):
        assert self.L.orc_enable_f32(self.h)

    def f32_images(self, n):
        p = self.L.orc_f32(self.h)
        return np.ctypeslib.as_array(p, shape=(n, IMG, IMG)).copy()

    def run(self, seed, env_id, n):
        return self.L.orc_run(self.h, seed, env_id, n)
