"""VecAssemblyGym: E independent assembly_gym environments advanced in lock-step
on one MI355X by the HIP kernels of libbridges_hip.so.

Host-side mirror of ``AssemblyGym`` + the per-step feature pipeline of
``rollout_episode`` (assembly_gym/assembly_gym/envs/gym_env.py:112-333,
robotoddler/training/successor_dqn.py:365-475).  All state lives in device
tensors allocated here (struct of arrays, see ``abi.ENV_BUFFER_FIELDS``); the
library gets raw pointers.  Nothing in here computes the simulation on the
host -- without the library or a GPU construction fails.

Lock-step protocol: after ``reset()`` every env holds the candidate
set of its (fresh) state.  ``step(sel)`` places candidate ``sel[e]`` of every
env (or performs a reset-only step for envs whose state had no valid
candidate), evaluates both stability variants, reward and termination,
auto-resets finished envs and produces the candidate set of the new state.
"""
import ctypes as C
import math
import os

import numpy as np
import torch

from . import abi
from .shapes import ShapeGeometry, load_urdf

DEFAULT_BOUNDS = ((-3.0, -3.0, -1.0), (7.0, 7.0, 9.0))      # assembly_env.py:168


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_stream = abi.current_stream


def gaussian_reward_map(target_image, kernel_size=101, sigma=16):
    """convolve_with_gaussian(targets raster, 101, 16) of get_task_features (successor_dqn.py:77-82, utils.py:93-115): the
    zero-padded 'same' cross-correlation of the 0/1 target image with the float32 kernel k k^T, k as the reference builds it
    (torch float32 on the host).  Evaluated on the host, in float64, set pixel by set pixel in row-major order, and rounded
    to float32 once: ONE fixed order on every box.  The library convolution this replaces (conv2d on the GPU) picked its
    algorithm from MIOpen's find results, which depend on the box and on what an earlier process left in MIOpen's cache --
    the same training run ended with different weights after an unrelated process had run convolutions on the box.
    target_image: numpy array [S, S] (the 0/1 targets raster; any weights work); returns float32 [S, S] (within 1e-7 relative
    of the library's float32 sum)."""
    coords = torch.arange(kernel_size) - kernel_size // 2
    k = torch.exp(-(coords.float() ** 2) / (2 * sigma ** 2))
    k = k / k.sum()
    k2 = (k.unsqueeze(0) * k.unsqueeze(1)).numpy().astype(np.float64)       # the float32 products, as the reference's kernel
    img = np.asarray(target_image, dtype=np.float64)
    S0, S1 = img.shape
    half = kernel_size // 2
    out = np.zeros((S0, S1), dtype=np.float64)
    ys, xs = np.arange(S0)[:, None], np.arange(S1)[None, :]
    for py, px in zip(*np.nonzero(img)):                                     # row-major
        i, j = py - ys + half, px - xs + half                                # out[y, x] += k2[py - y + half, px - x + half]
        ok = (i >= 0) & (i < kernel_size) & (j >= 0) & (j < kernel_size)
        out += img[py, px] * np.where(ok, k2[np.clip(i, 0, kernel_size - 1), np.clip(j, 0, kernel_size - 1)], 0.0)
    return out.astype(np.float32)


class ShapeTable:
    """Device copy of a list of shapes for the stand-alone operators."""

    def __init__(self, geoms):
        L = abi.require_gpu()
        self.geoms = list(geoms)
        arr = (abi.Shape * len(self.geoms))(*[g.to_struct() for g in self.geoms])
        self._dev = C.c_void_p()
        abi.check(L.bridges_shapes_upload(arr, len(self.geoms), C.byref(self._dev)), "bridges_shapes_upload")

    @property
    def ptr(self):
        return self._dev

    def __del__(self):
        try:
            if self._dev:
                abi.lib().bridges_shapes_free(self._dev)
        except Exception:
            pass


def raster_posed(table, verts, shape_ids, grid_x, grid_y, want_bits=True, want_f32=False):
    """bridges_raster_sized on n posed outlines (verts [n,6,2] f64, shape_ids [n] i32, device tensors); the image
    size S = len(grid_x) <= 64, outputs on the 64-word / 64x64 canvas (image = top-left S x S corner)."""
    L = abi.require_gpu()
    n = int(verts.shape[0])
    dev = verts.device
    size = int(grid_x.numel())
    if int(grid_y.numel()) != size:
        raise NotImplementedError("the HIP rasteriser renders square images")
    bits = torch.empty((n, 64), dtype=torch.int64, device=dev) if want_bits else None
    img = torch.empty((n, 64, 64), dtype=torch.float32, device=dev) if want_f32 else None
    abi.check(L.bridges_raster_sized(table.ptr, n, _ptr(verts), _ptr(shape_ids), _ptr(grid_x), _ptr(grid_y), size,
                                     _ptr(bits), _ptr(img), _stream()), "bridges_raster")
    return bits, img


def check_img_size(img_size):
    from .ops import image_size
    return image_size(img_size)


class VecAssemblyGym:
    def __init__(self, num_envs, shapes, obstacles, targets, max_steps=None, mu=0.8, density=1.0, bounds=None,
                 xlim=(-3.0, 7.0), ylim=(0.0, 10.0), x_discr_ground=None, offset_values=(0.0,), seed=0,
                 device="cuda:0", f32_rasters=True, a_max=None, img_size=(64, 64), debug=0, env_id_base=0,
                 sparse_raster_update=False, candidate_snapshots=True):
        L = abi.require_gpu()
        self.img = check_img_size(img_size)          # S; every image buffer stays a 64x64 canvas, see crop()
        self.L = L
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.E = int(num_envs)
        self.shapes = [s if isinstance(s, ShapeGeometry) else s.geometry for s in shapes]
        self.shape_target_faces = [list(getattr(s, "target_faces_2d", range(g.num_faces_2d)))
                                   for s, g in zip(shapes, self.shapes)]
        self.obstacles = [tuple(float(v) for v in o) for o in obstacles]
        self.targets = [tuple(float(v) for v in t) for t in targets]
        self.max_steps = int(max_steps) if max_steps else 0
        self.K = self.max_steps if self.max_steps else abi.MAX_BLOCKS
        if self.K > abi.MAX_BLOCKS:
            raise ValueError(f"max_steps {self.K} > {abi.MAX_BLOCKS} (BRIDGES_MAX_BLOCKS)")
        self.mu, self.density = float(mu), float(density)
        bounds = DEFAULT_BOUNDS if bounds is None else bounds
        self.bounds = tuple(tuple(float(v) for v in b) for b in bounds)
        self.xlim, self.ylim = tuple(map(float, xlim)), tuple(map(float, ylim))
        if x_discr_ground is None:
            x_discr_ground = np.linspace(-2, 0, 10)               # successor_dqn.py:611
        self.x_discr_ground = [float(v) for v in x_discr_ground]
        self.offset_values = [float(v) for v in offset_values]
        self.seed = int(seed)
        self.debug = int(debug)                     # kernel timing experiments only (bench.py --debug echoes it in its line)
        self.env_id_base = int(env_id_base)
        # the task's shape table = the env's shapes + cube06 for obstacles/targets (gym_env.py:277)
        self.cube06 = load_urdf("shapes/cube06.urdf")
        self.table_geoms = self.shapes + [self.cube06]
        self.groups = [(si, f) for si in range(len(self.shapes)) for f in self.shape_target_faces[si]]
        if len(self.groups) > abi.MAX_GROUPS:
            raise ValueError("too many (shape, target face) pairs")
        max_faces = max(g.num_faces_2d for g in self.shapes)
        bound = len(self.groups) * (len(self.x_discr_ground) + self.K * max_faces * len(self.offset_values))
        self.a_max = int(a_max) if a_max else bound
        self.f32_rasters = bool(f32_rasters)
        self.sparse_raster_update = bool(sparse_raster_update)
        # keep the "last block frozen" tableau of every env for candidate_stability_mask() (123 KB per env)
        self.candidate_snapshots = bool(candidate_snapshots)
        self.grid_x = np.linspace(self.xlim[0], self.xlim[1], self.img)    # rendering.py:108
        self.grid_y = np.linspace(self.ylim[1], self.ylim[0], self.img)
        self._alloc()
        self._task_features()
        self._create()
        self.reset()

    # ------------------------------------------------------------------ buffers
    def _alloc(self):
        E, K, Cc = self.E, self.K, self.E * self.a_max
        self.ws_stride = abi.ENV_LP_WS_DOUBLES            # per-env persistent tableau of the incremental simplex
        self.cand_ws_stride = abi.lp_ws_stride(K)
        dims = dict(E=E, K=K, C=Cc, E1=E + 1, IF=abi.MAX_INTERFACES, WS=self.ws_stride, CWS=abi.CAND_WS_SLOTS,
                    WSC=self.cand_ws_stride)
        self.buf = {}
        for name, dt, shape in abi.ENV_BUFFER_FIELDS:
            if name in ("cand_raster", "state_raster") and not self.f32_rasters:
                self.buf[name] = None
                continue
            if name in ("cand_raster_nz", "state_raster_nz") and not (self.f32_rasters and self.sparse_raster_update):
                self.buf[name] = None
                continue
            shp = tuple(dims[s] if s in dims else int(s) for s in shape.split(","))
            self.buf[name] = torch.zeros(shp, dtype=getattr(torch, dt), device=self.device)
        for name, dt, shape in abi.ENV_BUFFER_FIELDS_TAIL:
            shp = tuple(dims[s] if s in dims else int(s) for s in shape.split(","))
            self.buf[name] = torch.zeros(shp, dtype=getattr(torch, dt), device=self.device)
        self.stats = torch.zeros(16, dtype=torch.int64, device=self.device)
        self.lp_snap = (torch.zeros((E, abi.ENV_LP_SNAP_DOUBLES), dtype=torch.float64, device=self.device)
                        if self.candidate_snapshots else None)
        for k, v in self.buf.items():
            setattr(self, k, v)
        self._contacts_current = True

    def _task_features(self):
        """get_task_features (successor_dqn.py:67-85): obstacle raster and the Gaussian-blurred target raster,
        both rasterised by the HIP kernel from cube06 blocks (gym_env.py:277-284)."""
        dev = self.device
        self.table = ShapeTable(self.table_geoms)
        cube_id = len(self.table_geoms) - 1
        gx = torch.tensor(self.grid_x, dtype=torch.float64, device=dev)
        gy = torch.tensor(self.grid_y, dtype=torch.float64, device=dev)
        self.grid_x_dev, self.grid_y_dev = gx, gy

        def raster_points(points):
            if len(points) == 0:
                return torch.zeros(64, dtype=torch.int64, device=dev)
            verts = torch.zeros((len(points), 6, 2), dtype=torch.float64)
            for i, p in enumerate(points):              # Block(shape=cube06, position=p): identity rotation
                for k, (vx, vz) in enumerate(self.cube06.verts):
                    verts[i, k, 0] = p[0] + vx
                    verts[i, k, 1] = p[2] + vz
            ids = torch.full((len(points),), cube_id, dtype=torch.int32, device=dev)
            bits, _ = raster_posed(self.table, verts.to(dev), ids, gx, gy)
            off = torch.tensor([0, len(points)], dtype=torch.int32, device=dev)
            out = torch.empty(64, dtype=torch.int64, device=dev)
            abi.check(self.L.bridges_bits_or(1, _ptr(off), _ptr(bits), _ptr(out), _stream()), "bridges_bits_or")
            return out

        self.buf["obstacle_bits"].copy_(raster_points(self.obstacles))
        tbits = raster_points(self.targets)
        timg = torch.empty((1, 64, 64), dtype=torch.float32, device=dev)
        abi.check(self.L.bridges_bits_to_f32(1, _ptr(tbits), _ptr(timg), _stream()), "bridges_bits_to_f32")
        S = self.img                                                       # the map of the S x S image, rest of the canvas 0
        rm = np.zeros((64, 64), dtype=np.float32)
        rm[:S, :S] = gaussian_reward_map(timg[0, :S, :S].cpu().numpy())    # successor_dqn.py:80-82, once per task, on the host
        self.buf["reward_map"].copy_(torch.from_numpy(rm))
        # float64 row prefix sums of the map, accumulated left to right on the host (one fixed order on every box): the
        # rasteriser takes sum(raster * reward_map) of a candidate from the runs of its rows (bridges_env_buffers.reward_prefix)
        pre = np.zeros((64, 65), dtype=np.float64)
        pre[:, 1:] = np.cumsum(rm.astype(np.float64), axis=1)
        self.buf["reward_prefix"].copy_(torch.from_numpy(pre))
        oimg = torch.empty((1, 64, 64), dtype=torch.float32, device=dev)
        abi.check(self.L.bridges_bits_to_f32(1, _ptr(self.buf["obstacle_bits"]), _ptr(oimg), _stream()),
                  "bridges_bits_to_f32")
        self.obstacle_raster = self.crop(oimg)                            # [1,S,S] f32
        self.reward_features = self.crop(self.buf["reward_map"].unsqueeze(0))   # [1,S,S] f32
        self._reward_obstacle_flat = None                                        # (cache of the trainer: both maps, flattened)

    def crop(self, images):
        """[..., 64, 64] canvas -> the [..., S, S] image (a no-op view for the default S = 64)."""
        return images if self.img == 64 else images[..., :self.img, :self.img]

    def _create(self):
        t = abi.Task()
        t.n_envs, t.max_blocks, t.max_steps, t.a_max = self.E, self.K, self.max_steps, self.a_max
        t.n_shapes, t.n_groups = len(self.table_geoms), len(self.groups)
        for i, (si, f) in enumerate(self.groups):
            t.group_shape[i], t.group_face[i] = si, f
        t.n_ground, t.n_offsets, t.n_targets = len(self.x_discr_ground), len(self.offset_values), len(self.targets)
        t.mu, t.density = self.mu, self.density
        t.floor_half_width = (self.bounds[1][0] - self.bounds[0][0]) / 2.0      # assembly_env.py:290-296
        t.floor_depth = self.bounds[1][1] - self.bounds[0][1]
        self._create_args = (float(t.floor_half_width), float(t.floor_depth))
        t.xlim[0], t.xlim[1], t.ylim[0], t.ylim[1] = *self.xlim, *self.ylim
        if len(self.targets) > abi.MAX_TARGETS:
            raise ValueError("too many targets")
        for i, tg in enumerate(self.targets):
            for k in range(3):
                t.targets[i][k] = tg[k]
        t.seed = self.seed
        t.debug = self.debug
        t.env_id_base = self.env_id_base
        self._shape_arr = (abi.Shape * len(self.table_geoms))(*[g.to_struct() for g in self.table_geoms])
        self._xg = (C.c_double * len(self.x_discr_ground))(*self.x_discr_ground)
        self._off = (C.c_double * len(self.offset_values))(*self.offset_values)
        t.img_size = self.img
        self._gx = (C.c_double * self.img)(*self.grid_x.tolist())
        self._gy = (C.c_double * self.img)(*self.grid_y.tolist())
        t.shapes = C.cast(self._shape_arr, C.POINTER(abi.Shape))
        dp = C.POINTER(C.c_double)
        t.x_ground, t.offsets = C.cast(self._xg, dp), C.cast(self._off, dp)
        t.grid_x, t.grid_y = C.cast(self._gx, dp), C.cast(self._gy, dp)
        b = abi.EnvBuffers()
        for name, _, _ in abi.ENV_BUFFER_FIELDS:
            setattr(b, name, self.buf[name].data_ptr() if self.buf[name] is not None else None)
        b.lp_ws_stride = self.ws_stride
        b.stats = self.stats.data_ptr()
        for name, _, _ in abi.ENV_BUFFER_FIELDS_TAIL:
            setattr(b, name, self.buf[name].data_ptr())
        b.cand_ws_stride = self.cand_ws_stride
        b.lp_snap = self.lp_snap.data_ptr() if self.lp_snap is not None else None
        b.lp_snap_stride = abi.ENV_LP_SNAP_DOUBLES
        self._env = C.c_void_p()
        abi.check(self.L.bridges_env_create(C.byref(t), C.byref(b), C.byref(self._env)), "bridges_env_create")

    def __del__(self):
        try:
            if getattr(self, "_env", None):
                self.L.bridges_env_destroy(self._env)
        except Exception:
            pass

    # ------------------------------------------------------------------ lock-step API
    def reset(self):
        abi.check(self.L.bridges_env_reset(self._env, _stream()), "bridges_env_reset")
        self._contacts_current = True
        self._cand_version = getattr(self, "_cand_version", 0) + 1

    def select_random(self):
        """Synthetic uniform-random policy over each env's valid candidates -> sel_index."""
        abi.check(self.L.bridges_env_select_random(self._env, _stream()), "bridges_env_select_random")

    def step(self, sel_index=None):
        if sel_index is not None:
            self.buf["sel_index"].copy_(sel_index.to(device=self.device, dtype=torch.int32))
        abi.check(self.L.bridges_env_step(self._env, _stream()), "bridges_env_step")
        self._cand_version += 1

    def timing_begin(self, max_launches):
        abi.check(self.L.bridges_env_timing_begin(self._env, int(max_launches)), "bridges_env_timing_begin")

    def timing_end(self):
        """-> (total ms of the rasteriser launches, number of launches), measured with HIP events on the stream."""
        ms, n = C.c_double(0.0), C.c_int32(0)
        abi.check(self.L.bridges_env_timing_end(self._env, C.byref(ms), C.byref(n)), "bridges_env_timing_end")
        return ms.value, n.value

    def refresh(self):
        """Recompute the candidate set (enumerate, rasterise, mask) after the host edited the state arrays."""
        abi.check(self.L.bridges_env_refresh(self._env, _stream()), "bridges_env_refresh")
        self._cand_version += 1

    def load_states(self, n_blocks, blk_shape, blk_pose, blk_occ):
        """Overwrite the state of every env with caller-supplied block lists (replay re-rasterisation): world
        vertices, the state raster and the candidate set are rebuilt by the HIP operators."""
        E, K = self.E, self.K
        self.buf["n_blocks"].copy_(n_blocks)
        self.buf["blk_shape"].copy_(blk_shape)
        self.buf["blk_pose"].copy_(blk_pose)
        self.buf["blk_occ"].copy_(blk_occ)
        self.buf["needs_reset"].zero_()
        self.buf["n_if"].zero_()                      # interfaces are only needed by step(); replay states never step
        self._contacts_current = False
        flat_shape = self.buf["blk_shape"].reshape(E * K)
        abi.check(self.L.bridges_pose_block(self.table.ptr, E * K, _ptr(flat_shape), _ptr(self.buf["blk_pose"]),
                                            _ptr(self.buf["blk_verts"]), _stream()), "bridges_pose_block")
        bits = torch.empty((E * K, 64), dtype=torch.int64, device=self.device)
        abi.check(self.L.bridges_raster_sized(self.table.ptr, E * K, _ptr(self.buf["blk_verts"]), _ptr(flat_shape),
                                              _ptr(self.grid_x_dev), _ptr(self.grid_y_dev), self.img, _ptr(bits), None,
                                              _stream()), "bridges_raster")
        start = torch.arange(E, dtype=torch.int32, device=self.device) * K
        ranges = torch.stack([start, start + self.buf["n_blocks"]], dim=1).contiguous()
        abi.check(self.L.bridges_bits_or(E, _ptr(ranges), _ptr(bits), _ptr(self.buf["state_bits"]), _stream()),
                  "bridges_bits_or")
        self._keep = (bits, ranges, flat_shape)       # alive until the stream has consumed them
        # candidate counts of the loaded states, then the usual refresh
        nfree = torch.zeros(E, dtype=torch.int32, device=self.device)
        if getattr(self, "_nv_dev", None) is None:      # once: a host list -> device copy makes the host wait
            self._nv_dev = torch.tensor([g.num_faces_2d for g in self.table_geoms], dtype=torch.int32, device=self.device)
        nv = self._nv_dev
        kidx = torch.arange(K, device=self.device)[None, :]
        live = kidx < self.buf["n_blocks"][:, None]
        faces = nv[self.buf["blk_shape"].long()]
        occ = self.buf["blk_occ"].to(torch.int32)
        popc = sum(((occ >> f) & 1) for f in range(abi.MAX_VERTS))
        nfree = ((faces - popc) * live).sum(dim=1).to(torch.int32)
        n_cand = len(self.groups) * (len(self.x_discr_ground) + nfree * len(self.offset_values))
        self.buf["n_cand"].copy_(n_cand.to(torch.int32))       # raw count: refresh clamps to a_max and flags / counts a truncation
        self.refresh()

    def load_records(self, rec):
        """load_states for the next states s' of sampled transition records ([n <= E, RECORD_WIDTH] float64,
        robotoddler/training/records.py): ONE launch (bridges_replay_unpack) writes the block lists, the candidate
        counts and the block ranges instead of the ~60 slice / cast / index launches of unpack_states + load_states;
        envs beyond n repeat record 0.  Returns (bits of s, lin_reward f32, stable(s) f32, done u8, stable(s') u8), [E] each."""
        E, K, dev = self.E, self.K, self.device
        assert rec.dtype == torch.float64 and rec.is_contiguous() and 1 <= rec.shape[0] <= E
        if getattr(self, "_nv_dev", None) is None:
            self._nv_dev = torch.tensor([g.num_faces_2d for g in self.table_geoms], dtype=torch.int32, device=dev)
        ranges = torch.empty((2, E, 2), dtype=torch.int32, device=dev)
        lin, stable_s = torch.empty(E, dtype=torch.float32, device=dev), torch.empty(E, dtype=torch.float32, device=dev)
        done, stable_n = torch.empty(E, dtype=torch.uint8, device=dev), torch.empty(E, dtype=torch.uint8, device=dev)
        b = self.buf
        abi.check(self.L.bridges_replay_unpack(E, rec.shape[0], K, _ptr(rec), _ptr(self._nv_dev), self._nv_dev.numel(),
                                               len(self.groups), len(self.x_discr_ground), len(self.offset_values),
                                               _ptr(b["n_blocks"]), _ptr(b["blk_shape"]), _ptr(b["blk_pose"]), _ptr(b["blk_occ"]),
                                               _ptr(b["n_cand"]), _ptr(ranges[0]), _ptr(ranges[1]), _ptr(lin), _ptr(stable_s),
                                               _ptr(done), _ptr(stable_n), _stream()), "bridges_replay_unpack")
        b["needs_reset"].zero_()
        b["n_if"].zero_()                             # interfaces are only needed by step(); replay states never step
        self._contacts_current = False
        flat_shape = b["blk_shape"].reshape(E * K)
        abi.check(self.L.bridges_pose_block(self.table.ptr, E * K, _ptr(flat_shape), _ptr(b["blk_pose"]), _ptr(b["blk_verts"]),
                                            _stream()), "bridges_pose_block")
        bits = torch.empty((E * K, 64), dtype=torch.int64, device=dev)
        abi.check(self.L.bridges_raster_sized(self.table.ptr, E * K, _ptr(b["blk_verts"]), _ptr(flat_shape), _ptr(self.grid_x_dev),
                                              _ptr(self.grid_y_dev), self.img, _ptr(bits), None, _stream()), "bridges_raster")
        abi.check(self.L.bridges_bits_or(E, _ptr(ranges[0]), _ptr(bits), _ptr(b["state_bits"]), _stream()), "bridges_bits_or")
        bits_s = torch.empty((E, 64), dtype=torch.int64, device=dev)
        abi.check(self.L.bridges_bits_or(E, _ptr(ranges[1]), _ptr(bits), _ptr(bits_s), _stream()), "bridges_bits_or")
        self._keep = (bits, ranges[0], flat_shape)    # alive until the stream has consumed them
        self.refresh()
        return bits_s, lin, stable_s, done, stable_n

    def prefix_state_bits(self, n_prefix):
        """Bit raster of the first n_prefix[e] blocks of every env as loaded by the last load_states call (int32/int64
        [E]): the state a transition started from, when the env holds the state it led to."""
        bits, _, _ = self._keep
        start = torch.arange(self.E, dtype=torch.int32, device=self.device) * self.K
        ranges = torch.stack([start, start + n_prefix.to(torch.int32)], dim=1).contiguous()
        out = torch.empty((self.E, 64), dtype=torch.int64, device=self.device)
        abi.check(self.L.bridges_bits_or(self.E, _ptr(ranges), _ptr(bits), _ptr(out), _stream()), "bridges_bits_or")
        return out

    def candidate_stability_mask(self):
        """is_action_stable_rbe (assembly_gym/utils/stability.py:122-130 of the reference) for EVERY valid candidate of
        every env, fused on the device (bridges_env_candidate_stability): one wave per candidate appends the candidate
        block to its env's persistent contact list in LDS (the last placed block stays frozen, gym_env.py:238-240) and
        solves the feasibility LP.  Fills ``cand_stable`` (uint8 [C]: 1 stable, 0 unstable / masked-out, 2 solver
        error) without any host synchronisation; returns the number of decisions as a device scalar."""
        if not self._contacts_current:
            raise abi.BridgesHipError("candidate stability needs the persistent contact lists of states reached through "
                                      "reset()/step(); this env was overwritten by load_states()")
        abi.check(self.L.bridges_env_candidate_stability(self._env, _stream()), "bridges_env_candidate_stability")
        return self.buf["n_valid"].sum()

    def candidate_stability(self):
        """-> (rows, stable): compact indices of the valid candidates and a bool per row (errors count as unstable,
        stability.py:68 + gym_env.py:182)."""
        self.candidate_stability_mask()
        idx, _ = self.valid_rows()
        return idx, self.buf["cand_stable"][idx] == 1

    def state_groups(self, flag=None):
        """rep int32 [E]: the smallest env index that holds exactly this env's state -- block count and the shape, pose bits
        and face occupancy of its blocks, plus the caller's per-env ``flag`` byte (e.g. the 'stable' feature) -- found by a
        64-bit hash and verified word for word (bridges_env_groups; two launches, no wait).  Envs in the same state hold the
        same candidates in the same order, so valid_rows(rep) lets them share one set of rows."""
        hkey = getattr(self, "_hkey", None)
        if hkey is None:
            hkey = self._hkey = torch.empty(self.E, dtype=torch.int64, device=self.device)
        rep = torch.empty(self.E, dtype=torch.int32, device=self.device)
        if flag is not None:
            flag = flag.to(torch.uint8).contiguous()
        b = self.buf
        abi.check(self.L.bridges_env_groups(self.E, self.K, _ptr(b["n_blocks"]), _ptr(b["blk_shape"]), _ptr(b["blk_pose"]),
                                            _ptr(b["blk_occ"]), _ptr(flag), _ptr(hkey), _ptr(rep), _stream()), "bridges_env_groups")
        return rep

    def valid_rows(self, rep=None):
        """Compact indices of the valid (filtered) candidates and their owning env: the rows a Q-network is fed
        (bridges_valid_rows: two launches and ONE wait for the row count; torch.nonzero + gather were six launches and two
        waits).  With ``rep`` (state_groups) only the envs that represent their state get rows, and valid_segments() gives
        every env the range of its representative.  The returned tensors are views of two alternating buffers: they stay
        intact until the call after next."""
        cached = getattr(self, "_valid_rows", None)
        if cached is not None and cached[0] == self._cand_version and cached[5] is rep:     # same candidate set, same grouping
            return cached[1], cached[2]
        if getattr(self, "_vr_buf", None) is None:
            cap = self.buf["cand_mask"].numel()
            mk = lambda: (torch.empty(cap, dtype=torch.int64, device=self.device), torch.empty(cap, dtype=torch.int64, device=self.device),
                          torch.empty(self.E + 1, dtype=torch.int32, device=self.device),
                          torch.empty((2, self.E), dtype=torch.int32, device=self.device))
            self._vr_buf, self._vr_flip = (mk(), mk()), 0
            self._vr_total = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._vr_flip ^= 1
        idx_b, env_b, seg, lohi = self._vr_buf[self._vr_flip]
        b = self.buf
        abi.check(self.L.bridges_valid_rows(self.E, _ptr(b["cand_offset"]), _ptr(b["n_cand"]), _ptr(b["n_valid"]), _ptr(b["cand_mask"]),
                                            _ptr(rep), _ptr(seg), _ptr(lohi[0]) if rep is not None else None,
                                            _ptr(lohi[1]) if rep is not None else None, _ptr(idx_b), _ptr(env_b),
                                            C.c_void_p(self._vr_total.data_ptr()), _stream()), "bridges_valid_rows")
        torch.cuda.current_stream(self.device).synchronize()             # the one wait: the count is on the host now
        n = int(self._vr_total[0])
        idx, row_env = idx_b[:n], env_b[:n]
        segs = (lohi[0], lohi[1]) if rep is not None else (seg[:self.E], seg[1:])
        self._valid_rows = (self._cand_version, idx, row_env, seg, segs, rep)
        return idx, row_env

    def valid_segments(self):
        """Row ranges of the envs in the last ``valid_rows()``: (lo, hi), int32 [E] each -- rows lo[e] .. hi[e] are env e's
        (with a grouping: its representative's); without a grouping they are seg[:-1], seg[1:] of the prefix sums of n_valid."""
        if getattr(self, "_valid_rows", None) is None or self._valid_rows[0] != self._cand_version:
            self.valid_rows()
        return self._valid_rows[4]

    # ------------------------------------------------------------------ views
    def flags(self):
        f = self.buf["step_flags"]
        out = {n: f[:, i].bool() for i, n in enumerate(abi.FLAG_NAMES)}
        out["lp_error"] = (f[:, 7] & 3) != 0                    # bit 0 solver error, bit 1 contact-list overflow
        out["cand_overflow"] = (f[:, 7] & 8) != 0               # the state has more raw candidates than a_max: the set was cut
        out["warm_resolved"] = (f[:, 7] & 4) != 0               # a continued tableau's marginal 'unstable' was re-solved cold
        return out

    def binary_features(self):
        """get_state_features' binary vector [stable, collision x5] (successor_dqn.py:53-60).  The vector of the
        current state: 'stable' of a freshly reset env is True (empty assembly, stability.py:53-56)."""
        out = torch.zeros((self.E, 6), dtype=torch.float32, device=self.device)
        f = self.buf["step_flags"]
        fresh = self.buf["n_blocks"] == 0
        out[:, 0] = torch.where(fresh, torch.ones_like(fresh), f[:, 1].bool()).float()
        return out

    def total_candidates(self):
        return int(self.buf["cand_offset"][self.E].item())

    def read_stats(self):
        return dict(zip(abi.STAT_NAMES, self.stats.tolist()))


class VecAssemblyGymGroups:
    """E environments split into G independent groups, each a VecAssemblyGym on its own HIP stream.

    The lock-step of one group is a dependent chain (latency-bound wave-per-env task kernel, then the
    bandwidth-bound rasteriser); with two or more groups in flight the task kernel of one group overlaps the
    rasteriser of another.  Environments are independent, so results are identical to a single group with the same
    global env ids (policy RNG streams are keyed by seed and global env id)."""

    def __init__(self, num_envs, *args, groups=2, device="cuda:0", raster_gate=None, env_id_base=0, **kw):
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.G = int(groups)
        base = num_envs // self.G
        sizes = [base + (1 if g < num_envs % self.G else 0) for g in range(self.G)]
        self.E = int(num_envs)
        self.envs, self.streams, start = [], [], int(env_id_base)
        for n in sizes:
            st = torch.cuda.Stream(device=self.device)
            with torch.cuda.stream(st):
                self.envs.append(VecAssemblyGym(n, *args, device=device, env_id_base=start, **kw))
            self.streams.append(st)
            start += n
        self._stream_ptrs = [C.c_void_p(st.cuda_stream) for st in self.streams]
        # one raster gate per GPU: the bandwidth-bound rasterisers of the groups run one after another, the
        # latency-bound task kernels of the other groups run beside them
        self._gate = C.c_void_p()
        L = self.envs[0].L
        abi.check(L.bridges_gate_create(C.byref(self._gate)), "bridges_gate_create")
        if raster_gate is None:                     # the gate pays when the rasterisers are HBM-bound (full rewrite);
            # measured 3 groups, sparse row-group update: 6.89 M env-steps/s without it, 6.18 M with it
            raster_gate = not kw.get("sparse_raster_update")
        if self.G > 1 and raster_gate:
            for env in self.envs:
                abi.check(L.bridges_env_set_gate(env._env, self._gate), "bridges_env_set_gate")
        self.sync()

    def sync(self):
        for st in self.streams:
            st.synchronize()

    def __del__(self):
        try:
            self.sync()
            for env in self.envs:
                env.L.bridges_env_set_gate(env._env, None)
            self.envs[0].L.bridges_gate_destroy(self._gate)
        except Exception:
            pass

    def reset(self):
        for env, st in zip(self.envs, self.streams):
            with torch.cuda.stream(st):
                env.reset()

    def lockstep_random(self):
        """select_random + step for every group, each on its own stream (one C call per group)."""
        fn = self.envs[0].L.bridges_env_lockstep_random
        for env, sp in zip(self.envs, self._stream_ptrs):
            rc = fn(env._env, sp)
            env._cand_version += 1
            if rc != 0:
                abi.check(rc, "bridges_env_lockstep_random")

    def lockstep_random_candidates(self, timed=None):
        """lockstep_random, then is_action_stable_rbe for every valid candidate of the new states (candidate_stability_mask),
        group by group on the groups' own streams: the latency-bound LP pass of one group runs beside the bandwidth-bound
        rasteriser of the next.  The envs must have been created with candidate_snapshots=True.  ``timed``: a list that
        receives (start event, end event, decisions as a device scalar) of every group's LP pass."""
        fn = self.envs[0].L.bridges_env_lockstep_random
        for env, st, sp in zip(self.envs, self.streams, self._stream_ptrs):
            with torch.cuda.stream(st):
                rc = fn(env._env, sp)
                env._cand_version += 1
                if rc != 0:
                    abi.check(rc, "bridges_env_lockstep_random")
                if timed is not None:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    n = env.candidate_stability_mask()
                    b.record()
                    timed.append((a, b, n))
                else:
                    env.candidate_stability_mask()

    def timing_begin(self, max_launches):
        for env in self.envs:
            env.timing_begin(max_launches)

    def timing_end(self):
        ms = n = 0
        for env in self.envs:
            a, b = env.timing_end()
            ms += a
            n += b
        return ms, n

    def read_stats(self):
        out = {}
        for env in self.envs:
            for k, v in env.read_stats().items():
                out[k] = out.get(k, 0) + v
        return out
