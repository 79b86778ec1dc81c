"""Thin host wrappers of the stand-alone HIP operators (single calls on caller-shaped batches).

Used by the single-environment ``assembly_gym`` API; every function launches a kernel of
libbridges_hip.so on the current torch stream and returns device tensors."""
import ctypes as C

import numpy as np
import torch

from . import abi
from .shapes import ShapeGeometry

_tables = {}


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_stream = abi.current_stream


class ShapeRegistry:
    """Process-wide device table of every ShapeGeometry seen so far (ids are stable)."""

    def __init__(self):
        self.geoms = []
        self._dev = None
        self._dev_count = 0

    def id_of(self, geom: ShapeGeometry):
        for i, g in enumerate(self.geoms):
            if g is geom:
                return i
        self.geoms.append(geom)
        return len(self.geoms) - 1

    def device_table(self):
        L = abi.require_gpu()
        if self._dev is None or self._dev_count != len(self.geoms):
            if self._dev is not None:
                L.bridges_shapes_free(self._dev)
            arr = (abi.Shape * len(self.geoms))(*[g.to_struct() for g in self.geoms])
            dev = C.c_void_p()
            abi.check(L.bridges_shapes_upload(arr, len(self.geoms), C.byref(dev)), "bridges_shapes_upload")
            self._dev, self._dev_count = dev, len(self.geoms)
        return self._dev


REGISTRY = ShapeRegistry()


class _TorchDtypes(dict):
    def __missing__(self, dt):                              # any other numpy dtype torch has a name for
        self[dt] = getattr(torch, np.dtype(dt).name)
        return self[dt]


_TORCH_DTYPE = _TorchDtypes({np.dtype(n): getattr(torch, n) for n in ("float32", "float64", "int32", "int64", "uint8", "int8", "int16", "bool")})


class _Stager:
    """Small host arrays -> device tensors through ONE pinned staging buffer and ONE asynchronous copy per call.

    ``torch.tensor(array, device=...)`` of pageable memory is a blocking copy: the host waits for everything queued on the
    stream before it.  An operator of the single-environment API takes three to seven such arrays (stability: poses,
    vertices, shape ids, count, frozen mask), i.e. as many host waits per call -- 30 of the ~50 waits of one env-step of the
    reference-style loop.  Here the arrays are packed (16-byte aligned) into one of a ring of pinned buffers, copied with one
    non-blocking transfer, and handed out as typed views of one device buffer; a slot is reused only after the event
    recorded behind its copy has passed."""
    SLOTS, MIN_BYTES = 16, 1 << 16

    def __init__(self):
        self.slots, self.turn = [], 0

    def upload(self, dev, *arrays):
        arrays = [a if a.flags.c_contiguous else np.ascontiguousarray(a) for a in arrays]
        offs, total = [], 0
        for a in arrays:
            offs.append(total)
            total += -(-max(a.nbytes, 1) // 16) * 16
        if len(self.slots) < self.SLOTS:
            pinned = torch.empty(max(total, self.MIN_BYTES), dtype=torch.uint8).pin_memory()
            self.slots.append([pinned, torch.cuda.Event(), pinned.numpy()])     # buffer, "copy done" event, numpy view
            slot = self.slots[-1]
        else:
            slot = self.slots[self.turn]
            self.turn = (self.turn + 1) % self.SLOTS
            slot[1].synchronize()                           # (passed long ago unless 16 uploads are in flight)
            if slot[0].numel() < total:
                slot[0] = torch.empty(total, dtype=torch.uint8).pin_memory()
                slot[2] = slot[0].numpy()
        host = slot[2]
        for a, off in zip(arrays, offs):
            host[off:off + a.nbytes] = a.reshape(-1).view(np.uint8)
        d = torch.empty(total, dtype=torch.uint8, device=dev)
        d.copy_(slot[0][:total], non_blocking=True)
        slot[1].record()
        if len(arrays) == 1:
            a = arrays[0]
            return [d[:a.nbytes].view(_TORCH_DTYPE[a.dtype]).reshape(a.shape)]
        return [d[off:off + a.nbytes].view(_TORCH_DTYPE[a.dtype]).reshape(a.shape) for a, off in zip(arrays, offs)]


_STAGER = _Stager()


def upload(dev, *arrays):
    """numpy arrays (any dtypes) -> device tensors of the same shapes, one pinned staging copy for all of them."""
    return _STAGER.upload(dev, *arrays)


def download(*tensors):
    """Device tensors -> numpy arrays with ONE device-to-host copy (one host wait): they are packed into a float64 buffer on
    the device first (integer and byte tensors are exact in float64 here: counts, flags, small ids)."""
    flat = torch.cat([t.reshape(-1).to(torch.float64) for t in tensors])
    host = flat.cpu().numpy()
    out, off = [], 0
    for t in tensors:
        n = t.numel()
        a = host[off:off + n].reshape(tuple(t.shape))
        off += n
        out.append(a if t.dtype == torch.float64 else a.astype(np.dtype(str(t.dtype).replace("torch.", ""))))
    return out


_grids = {}


def pixel_grid(dev, xlim, ylim, S):
    """(grid_x, grid_y) of render_blocks_2d on the device, cached per (device, limits, size)."""
    key = (str(dev), float(xlim[0]), float(xlim[1]), float(ylim[0]), float(ylim[1]), int(S))
    g = _grids.get(key)
    if g is None:
        if len(_grids) > 64:
            _grids.clear()
        g = _grids[key] = tuple(upload(dev, np.linspace(xlim[0], xlim[1], S), np.linspace(ylim[1], ylim[0], S)))
    return g


def device():
    abi.require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


def place(frame1, geom, face, ox, oy):
    """create_block / align_frames_2d (gym_env.py:204-216, geometry.py:39-50) for ONE placement.
    frame1 = (cx, cz, tx, tz, nx, nz).  Returns (pose[4], verts[nv,2]) as float64 numpy arrays."""
    L = abi.require_gpu()
    dev = device()
    sid = REGISTRY.id_of(geom)
    tab = REGISTRY.device_table()
    f1, sh, fc, oxs, oys = upload(dev, np.array([list(frame1)], dtype=np.float64), np.array([sid], dtype=np.int32),
                                  np.array([int(face)], dtype=np.int32), np.array([float(ox)]), np.array([float(oy)]))
    pose = torch.empty((1, 4), dtype=torch.float64, device=dev)
    verts = torch.empty((1, 6, 2), dtype=torch.float64, device=dev)
    abi.check(L.bridges_place(tab, 1, _ptr(f1), _ptr(sh), _ptr(fc), _ptr(oxs), _ptr(oys), _ptr(pose), _ptr(verts),
                              _stream()), "bridges_place")
    pose_h, verts_h = download(pose, verts)
    return pose_h[0], verts_h[0, :len(geom.verts)]


def image_size(img_size):
    """-> S of an S x S image the rasteriser renders (2 <= S <= 64, one wave lane per pixel column)."""
    w, h = (int(v) for v in img_size)
    if w != h or not 2 <= w <= 64:
        raise NotImplementedError(f"the HIP rasteriser renders square images of 2..64 pixels, not {w}x{h} "
                                  "(successor_dqn.py:585 default: 64x64)")
    return w


def crop(images, img_size):
    """[..., 64, 64] canvas -> the [..., S, S] image in its top-left corner (no-op for 64x64)."""
    S = image_size(img_size)
    return images if S == 64 else images[..., :S, :S]


def raster_bits(blocks, xlim, ylim, img_size=(64, 64)):
    """One bit raster per posed block: int64 [n,64] on the device (rendering.py:105-113 per block).  For S x S
    images with S < 64 the words / bits >= S are zero."""
    S = image_size(img_size)
    L = abi.require_gpu()
    dev = device()
    n = len(blocks)
    if n == 0:
        return torch.zeros((0, 64), dtype=torch.int64, device=dev)
    verts = np.zeros((n, 6, 2))
    ids = np.zeros(n, dtype=np.int32)
    for i, b in enumerate(blocks):
        verts[i, :len(b.verts_2d)] = b.verts_2d
        ids[i] = REGISTRY.id_of(b.geometry)
    tab = REGISTRY.device_table()
    v, s = upload(dev, verts, ids)
    gx, gy = pixel_grid(dev, xlim, ylim, S)
    bits = torch.empty((n, 64), dtype=torch.int64, device=dev)
    abi.check(L.bridges_raster_sized(tab, n, _ptr(v), _ptr(s), _ptr(gx), _ptr(gy), S, _ptr(bits), None, _stream()),
              "bridges_raster")
    return bits


def render_blocks(blocks, xlim, ylim, img_size):
    """render_blocks_2d at any image size (bridges_render_blocks): the union of the posed blocks on the grid
    X = linspace(xlim, img_size[0]), Y = linspace(ylim[1], ylim[0], img_size[1]) -> uint8 [img_size[1], img_size[0]] (rows
    top to bottom) on the device."""
    L = abi.require_gpu()
    dev = device()
    W, H = int(img_size[0]), int(img_size[1])
    if W < 1 or H < 1:
        raise ValueError(f"image size {img_size}")
    n = len(blocks)
    verts = np.zeros((max(n, 1), 6, 2))
    ids = np.zeros(max(n, 1), dtype=np.int32)
    for i, b in enumerate(blocks):
        verts[i, :len(b.verts_2d)] = b.verts_2d
        ids[i] = REGISTRY.id_of(b.geometry)
    tab = REGISTRY.device_table()
    v, s, gx, gy = upload(dev, verts, ids, np.linspace(xlim[0], xlim[1], W), np.linspace(ylim[1], ylim[0], H))
    out = torch.empty((H, W), dtype=torch.uint8, device=dev)
    abi.check(L.bridges_render_blocks(tab, n, _ptr(v), _ptr(s), _ptr(gx), W, _ptr(gy), H, _ptr(out), _stream()), "bridges_render_blocks")
    return out


def reward_prefix(reward_map):
    """float64 row prefix sums [64, 65] of a reward map ([64,64] f32 tensor), accumulated left to right on the host: what
    bridges_action_features takes the linear reward of a candidate from (build it once per task)."""
    pre = np.zeros((64, 65), dtype=np.float64)
    pre[:, 1:] = np.cumsum(reward_map.detach().cpu().numpy().astype(np.float64).reshape(64, 64), axis=1)
    return upload(reward_map.device, pre)[0]


def action_features(blocks, xlim, ylim, state_bits=None, obstacle_bits=None, reward_map=None, img_size=(64, 64), want_f32=False,
                    prefix=None):
    """bridges_action_features for a list of posed candidate blocks against one state: -> (bits [n,64] int64, img
    [n,64,64] f32 or None, mask [n] bool, lin [n] f32 or None) on the device.  state_bits / obstacle_bits: [64] int64 bit
    rasters (None = empty); reward_map: [64,64] f32 tensor (None: no linear reward) or ``prefix`` = reward_prefix(map),
    built once."""
    S = image_size(img_size)
    L = abi.require_gpu()
    dev = device()
    n = len(blocks)
    bits = torch.empty((n, 64), dtype=torch.int64, device=dev)
    img = torch.empty((n, 64, 64), dtype=torch.float32, device=dev) if want_f32 else None
    mask = torch.zeros(n, dtype=torch.uint8, device=dev)
    lin = torch.zeros(n, dtype=torch.float32, device=dev) if (reward_map is not None or prefix is not None) else None
    if n == 0:
        return bits, img, mask.bool(), lin
    verts = np.zeros((n, 6, 2))
    ids = np.zeros(n, dtype=np.int32)
    for i, b in enumerate(blocks):
        verts[i, :len(b.verts_2d)] = b.verts_2d
        ids[i] = REGISTRY.id_of(b.geometry)
    tab = REGISTRY.device_table()
    v, s_ = upload(dev, verts, ids)
    gx, gy = pixel_grid(dev, xlim, ylim, S)
    if prefix is None and reward_map is not None:
        prefix = reward_prefix(reward_map)
    abi.check(L.bridges_action_features(tab, n, _ptr(v), _ptr(s_), _ptr(gx), _ptr(gy), S, float(xlim[0]), float(xlim[1]),
                                        float(ylim[0]), float(ylim[1]), _ptr(state_bits), _ptr(obstacle_bits), _ptr(prefix),
                                        _ptr(bits), _ptr(img), _ptr(mask), _ptr(lin), _stream()), "bridges_action_features")
    return bits, img, mask.bool(), lin


def bits_or(bits):
    L = abi.require_gpu()
    dev = bits.device
    out = torch.zeros(64, dtype=torch.int64, device=dev)
    if bits.shape[0] == 0:
        return out
    off, = upload(dev, np.array([0, bits.shape[0]], dtype=np.int32))
    abi.check(L.bridges_bits_or(1, _ptr(off), _ptr(bits), _ptr(out), _stream()), "bridges_bits_or")
    return out


def bits_to_f32(bits):
    """[n,64] int64 -> [n,64,64] float32 {0,1}."""
    L = abi.require_gpu()
    bits = bits.reshape(-1, 64).contiguous()
    img = torch.empty((bits.shape[0], 64, 64), dtype=torch.float32, device=bits.device)
    if bits.shape[0]:
        abi.check(L.bridges_bits_to_f32(bits.shape[0], _ptr(bits), _ptr(img), _stream()), "bridges_bits_to_f32")
    return img


def bits_linear(bits, wt, bits_row=None, base=None, base_row=None):
    """out[r] = base[base_row[r]] + sum of the rows of ``wt`` ([4096, d], pixel-major) selected by the set pixels of the
    bit-packed raster ``bits[bits_row[r]]`` ([*,64] int64): a linear layer applied to flattened binary images without
    expanding them (bridges_bits_linear).  Returns [n, d] float32."""
    L = abi.require_gpu()
    bits = bits.reshape(-1, 64)
    assert bits.is_contiguous() and bits.dtype == torch.int64
    wt = wt.to(torch.float32).contiguous()
    assert wt.shape[0] == 4096 and wt.shape[1] % 4 == 0, "wt must be [4096, d] with d % 4 == 0"
    d = wt.shape[1]
    n = bits_row.numel() if bits_row is not None else bits.shape[0]
    if bits_row is not None:
        bits_row = bits_row.to(torch.int64).contiguous()
    if base is not None:
        base = base.to(torch.float32).reshape(-1, d).contiguous()
        if base_row is not None:
            base_row = base_row.to(torch.int64).contiguous()
            assert base_row.numel() == n
    out = torch.empty((n, d), dtype=torch.float32, device=bits.device)
    abi.check(L.bridges_bits_linear(n, _ptr(bits), _ptr(bits_row), _ptr(wt), d, _ptr(base), _ptr(base_row), _ptr(out),
                                    _stream()), "bridges_bits_linear")
    return out


def _head_splits(n_rows, n_tiles, slots=512):
    """Column ranges per 128-row workgroup: enough workgroups that the chip's 2 x 256 resident slots stay evenly filled to the
    end of the launch, few enough that the 128-KB row slab each one loads first stays small against its tiles (measured on
    45 k rows x 128 tiles: 1: 1.12 ms, 2: 0.87, 4: 0.86, 8: 0.81, 16: 0.85, 32: 0.94)."""
    wgs = (n_rows + 127) // 128
    best, best_cost = 1, None
    for s in (1, 2, 4, 8, 16, 32):
        if s > n_tiles:
            break
        per = -(-n_tiles // s)
        used = -(-n_tiles // per)
        unit = per + 2.0                                    # tile-times of one workgroup: its tiles + the slab load
        cost = max(wgs * used / slots, 1.0) * unit + unit   # the even part + the tail of the last workgroups
        if best_cost is None or cost < best_cost:
            best, best_cost = s, cost
    return best


def head_sigmoid_dot(h, Wd, bd, w, splits=None):
    """out[r] = sum_j w[j] * sigmoid(h[r] . Wd[j] + bd[j]) without materialising the [n, N] product (bridges_head_sigmoid_dot):
    h [n, 256] float32 (rows contiguous), Wd [N, 256], bd [N], w [N]."""
    L = abi.require_gpu()
    assert h.dtype == torch.float32 and h.dim() == 2 and h.stride(1) == 1 and h.shape[1] == 256
    Wd, bd, w = Wd.to(torch.float32).contiguous(), bd.to(torch.float32).contiguous(), w.to(torch.float32).reshape(-1).contiguous()
    N = Wd.shape[0]
    assert Wd.shape[1] == 256 and bd.numel() == N and w.numel() == N
    n = h.shape[0]
    if splits is None:
        splits = _head_splits(n, (N + 31) // 32)
    out = torch.empty(n, dtype=torch.float32, device=h.device)
    part = torch.empty((splits, n), dtype=torch.float32, device=h.device) if splits > 1 else None
    abi.check(L.bridges_head_sigmoid_dot(n, 256, N, _ptr(h), h.stride(0), _ptr(Wd), _ptr(bd), _ptr(w), _ptr(out),
                                         _ptr(part) if part is not None else None, splits, _stream()), "bridges_head_sigmoid_dot")
    return out


def eps_greedy_select(seg, q, join, u, eps, greedy, idx, cand_offset, rep=None):
    """EpsilonGreedy.select for every env in one launch (bridges_eps_greedy_select): seg int32 [E + 1] row ranges -- or a pair
    (lo, hi) of int32 [E] tensors, with rep int32 [E] = the env whose candidates the rows of env e index (envs in the same
    state share rows, bridges_env_groups) --, q / join float32 [n] (n >= 1), u float32 [E] uniform draws, idx int64 [n] compact
    candidate index of every row, cand_offset int32 [E].
    -> (sel_compact int64 [E], sel_index int32 [E], q_sel float32 [E], explore_w float32 [E])."""
    L = abi.require_gpu()
    if isinstance(seg, (tuple, list)):
        seg_lo, seg_hi = seg
        E = seg_lo.numel()
    else:
        E = seg.numel() - 1
        seg_lo, seg_hi = seg[:E], seg[1:]
    n = q.numel()
    assert seg_lo.dtype == torch.int32 and seg_hi.dtype == torch.int32 and seg_lo.is_contiguous() and seg_hi.is_contiguous()
    assert idx.dtype == torch.int64 and cand_offset.dtype == torch.int32 and n >= 1
    assert rep is None or (rep.dtype == torch.int32 and rep.numel() == E and rep.is_contiguous())
    q, join, u = q.to(torch.float32).contiguous(), join.to(torch.float32).contiguous(), u.to(torch.float32).contiguous()
    idx, cand_offset = idx.contiguous(), cand_offset.contiguous()
    assert join.numel() == n and idx.numel() == n and u.numel() == E and cand_offset.numel() >= E
    dev = q.device
    sel_compact = torch.empty(E, dtype=torch.int64, device=dev)
    sel_index = torch.empty(E, dtype=torch.int32, device=dev)
    q_sel, explore_w = torch.empty(E, dtype=torch.float32, device=dev), torch.empty(E, dtype=torch.float32, device=dev)
    abi.check(L.bridges_eps_greedy_select(E, n, _ptr(seg_lo), _ptr(seg_hi), _ptr(q), _ptr(join), _ptr(u), float(eps), int(bool(greedy)),
                                          _ptr(idx), _ptr(cand_offset), _ptr(rep), _ptr(sel_compact), _ptr(sel_index), _ptr(q_sel), _ptr(explore_w), _stream()),
              "bridges_eps_greedy_select")
    return sel_compact, sel_index, q_sel, explore_w


def bits_dot(bits, img, slot, bits_row=None):
    """out[r] = sum(img[slot[r]] * raster(bits[bits_row[r]])) for bit-packed 64x64 rasters (bridges_bits_dot): img
    [n_slots,64,64] float32, slot [n] int64 -> [n] float32."""
    L = abi.require_gpu()
    bits = bits.reshape(-1, 64)
    assert bits.is_contiguous() and bits.dtype == torch.int64 and img.dtype == torch.float32 and img.is_contiguous()
    assert tuple(img.shape[-2:]) == (64, 64)
    slot = slot.to(torch.int64).contiguous()
    n = slot.numel()
    if bits_row is not None:
        bits_row = bits_row.to(torch.int64).contiguous()
        assert bits_row.numel() == n
    out = torch.empty(n, dtype=torch.float32, device=bits.device)
    abi.check(L.bridges_bits_dot(n, _ptr(bits), _ptr(bits_row), _ptr(img), _ptr(slot), _ptr(out), _stream()), "bridges_bits_dot")
    return out


def bits_accumulate_(img, bits, slot, weight=None, bits_row=None):
    """img[slot[r]] += weight[r] * raster(bits[bits_row[r]]) in place (bridges_bits_accumulate)."""
    L = abi.require_gpu()
    bits = bits.reshape(-1, 64)
    assert bits.is_contiguous() and bits.dtype == torch.int64 and img.dtype == torch.float32 and img.is_contiguous()
    assert tuple(img.shape[-2:]) == (64, 64)
    slot = slot.to(torch.int64).contiguous()
    n = slot.numel()
    if bits_row is not None:
        bits_row = bits_row.to(torch.int64).contiguous()
    if weight is not None:
        weight = weight.to(torch.float32).contiguous()
    abi.check(L.bridges_bits_accumulate(n, _ptr(bits), _ptr(bits_row), _ptr(weight), _ptr(slot), _ptr(img), _stream()),
              "bridges_bits_accumulate")
    return img


def sigmoid_dot(d, w):
    """out[r] = sum_j w[j] * sigmoid(d[r, j]) in one pass over ``d`` ([n, k] float32, k % 4 == 0; bridges_sigmoid_dot)."""
    L = abi.require_gpu()
    assert d.dtype == torch.float32 and d.dim() == 2 and d.stride(1) == 1
    w = w.to(torch.float32).reshape(-1).contiguous()
    n, k = d.shape
    assert w.numel() == k
    out = torch.empty(n, dtype=torch.float32, device=d.device)
    abi.check(L.bridges_sigmoid_dot(n, _ptr(d), d.stride(0), _ptr(w), k, _ptr(out), _stream()), "bridges_sigmoid_dot")
    return out


def stability(blocks, fixed, mu, density, floor_half_width, floor_depth, tension_tol=None):
    """is_stable_rbe (stability.py:49-71) of ONE assembly.  Returns (stable: bool, info: dict).
    With ``tension_tol`` the penalty variant (is_stable_rbe_penalty, stability.py:75-88): contact points may pull, stable
    iff an equilibrium with total tension <= tension_tol exists; info['forces'] then holds that equilibrium's contact
    forces [n_interfaces, 2, 3] = (compression, tension, tangential) per contact point."""
    L = abi.require_gpu()
    dev = device()
    K = abi.MAX_BLOCKS
    n = len(blocks)
    if n > K:
        raise abi.BridgesHipError(f"{n} blocks > BRIDGES_MAX_BLOCKS ({K})")
    if n == 0:                                   # empty assembly: no edges, no free node (stability.py:53-56)
        out = dict(objective=0.0, n_interfaces=0, pivots=0)
        if tension_tol is not None:
            out["forces"] = np.zeros((0, 2, 3))
        return True, out
    pose = np.zeros((1, K, 4))
    verts = np.zeros((1, K, 6, 2))
    ids = np.zeros((1, K), dtype=np.int32)
    mask = 0
    for i, b in enumerate(blocks):
        pose[0, i] = b.pose
        verts[0, i, :len(b.verts_2d)] = b.verts_2d
        ids[0, i] = REGISTRY.id_of(b.geometry)
        if i in fixed:
            mask |= 1 << i
    tab = REGISTRY.device_table()
    ws_stride = abi.lp_ws_stride(K)
    ws = torch.empty((1, ws_stride), dtype=torch.float64, device=dev)
    stable = torch.zeros(1, dtype=torch.uint8, device=dev)
    info = torch.zeros((1, 8), dtype=torch.float64, device=dev)
    # one staging copy for all five inputs; the tensors stay alive until the call returned
    a_pose, a_verts, a_ids, a_n, a_mask = upload(dev, pose, verts, ids, np.array([n], dtype=np.int32), np.array([mask], dtype=np.int32))
    forces = None
    if tension_tol is None:
        abi.check(L.bridges_stability(tab, 1, K, _ptr(a_pose), _ptr(a_verts), _ptr(a_ids), _ptr(a_n), _ptr(a_mask),
                                      float(mu), float(density), float(floor_half_width), float(floor_depth),
                                      _ptr(stable), _ptr(info), _ptr(ws), ws_stride, _stream()), "bridges_stability")
    else:
        forces = torch.zeros((1, abi.MAX_INTERFACES, 2, 3), dtype=torch.float64, device=dev)
        abi.check(L.bridges_stability_penalty(tab, 1, K, _ptr(a_pose), _ptr(a_verts), _ptr(a_ids), _ptr(a_n), _ptr(a_mask),
                                              float(mu), float(density), float(floor_half_width), float(floor_depth),
                                              float(tension_tol), _ptr(stable), _ptr(info), _ptr(forces), _ptr(ws), ws_stride,
                                              _stream()), "bridges_stability_penalty")
    if forces is None:
        inf, st = download(info[0], stable)                   # verdict and diagnostics in ONE copy back
    else:
        inf, st, fo = download(info[0], stable, forces[0])
    if inf[3] != 0:
        return None, dict(error="lp", objective=float(inf[0]), n_interfaces=int(inf[1]), pivots=int(inf[2]))
    out = dict(objective=float(inf[0]), n_interfaces=int(inf[1]), pivots=int(inf[2]))
    if forces is not None:
        out["forces"] = fo[:int(inf[1])]
    return bool(st[0]), out


def stability_variants(blocks, fixed_sets, mu, density, floor_half_width, floor_depth):
    """is_stable_rbe of the SAME blocks under several frozen sets, in one operator call and one copy back: a list of
    (stable, info) as ``stability`` returns them.  (AssemblyGym.step asks for the new block frozen and, right after, for it
    free: gym_env.py:238-243 with :325-333 of the reference.)"""
    L = abi.require_gpu()
    dev = device()
    K = abi.MAX_BLOCKS
    n, m = len(blocks), len(fixed_sets)
    if n > K:
        raise abi.BridgesHipError(f"{n} blocks > BRIDGES_MAX_BLOCKS ({K})")
    if n == 0:
        return [(True, dict(objective=0.0, n_interfaces=0, pivots=0)) for _ in fixed_sets]
    pose = np.zeros((m, K, 4))
    verts = np.zeros((m, K, 6, 2))
    ids = np.zeros((m, K), dtype=np.int32)
    for i, b in enumerate(blocks):
        pose[:, i] = b.pose
        verts[:, i, :len(b.verts_2d)] = b.verts_2d
        ids[:, i] = REGISTRY.id_of(b.geometry)
    masks = np.array([sum(1 << i for i in fixed if i < n) for fixed in fixed_sets], dtype=np.int32)
    tab = REGISTRY.device_table()
    ws_stride = abi.lp_ws_stride(K)
    ws = torch.empty((m, ws_stride), dtype=torch.float64, device=dev)
    stable = torch.zeros(m, dtype=torch.uint8, device=dev)
    info = torch.zeros((m, 8), dtype=torch.float64, device=dev)
    a_pose, a_verts, a_ids, a_n, a_mask = upload(dev, pose, verts, ids, np.full(m, n, dtype=np.int32), masks)
    abi.check(L.bridges_stability(tab, m, K, _ptr(a_pose), _ptr(a_verts), _ptr(a_ids), _ptr(a_n), _ptr(a_mask),
                                  float(mu), float(density), float(floor_half_width), float(floor_depth),
                                  _ptr(stable), _ptr(info), _ptr(ws), ws_stride, _stream()), "bridges_stability")
    inf, st = download(info, stable)
    out = []
    for j in range(m):
        d = dict(objective=float(inf[j, 0]), n_interfaces=int(inf[j, 1]), pivots=int(inf[j, 2]))
        out.append((None, dict(d, error="lp")) if inf[j, 3] != 0 else (bool(st[j]), d))
    return out


def create_blocks(target_blocks, target_faces, geoms, faces, oxs, oys):
    """AssemblyGym.create_block for n candidate placements in ONE bridges_create_block + ONE bridges_face_frames launch
    and three device-to-host copies (the per-candidate form costs a launch pair and three blocking copies each).
    target_blocks[i] = None for the floor, else an object with ``verts_2d`` / ``geometry``.  Returns per placement
    (pose[4], verts[nv,2], frames[nv,6]) float64 numpy -- the same values as n single calls (the kernels work per item)."""
    L = abi.require_gpu()
    dev = device()
    n = len(geoms)
    if n == 0:
        return []
    sid = np.array([REGISTRY.id_of(g) for g in geoms], dtype=np.int32)
    tsid = np.array([REGISTRY.id_of(t.geometry) if t is not None else sid[i] for i, t in enumerate(target_blocks)], dtype=np.int32)
    tab = REGISTRY.device_table()
    tv = np.zeros((n, 6, 2))
    tf = np.full(n, -1, dtype=np.int32)
    for i, t in enumerate(target_blocks):
        if t is not None:
            tv[i, :len(t.verts_2d)] = t.verts_2d
            tf[i] = int(target_faces[i])
    pose = torch.empty((n, 4), dtype=torch.float64, device=dev)
    verts = torch.empty((n, 6, 2), dtype=torch.float64, device=dev)
    frames = torch.empty((n, 6, 6), dtype=torch.float64, device=dev)
    sh, a_tv, a_ts, a_tf, a_face, a_ox, a_oy = upload(dev, sid, tv, tsid, tf, np.asarray(faces, dtype=np.int32),
                                                      np.asarray(oxs, dtype=np.float64), np.asarray(oys, dtype=np.float64))
    abi.check(L.bridges_create_block(tab, n, _ptr(a_tv), _ptr(a_ts), _ptr(a_tf), _ptr(sh), _ptr(a_face), _ptr(a_ox),
                                     _ptr(a_oy), _ptr(pose), _ptr(verts), None, _stream()), "bridges_create_block")
    abi.check(L.bridges_face_frames(tab, n, _ptr(sh), _ptr(verts), _ptr(frames), _stream()), "bridges_face_frames")
    pose_h, verts_h, frames_h = download(pose, verts, frames)
    return [(pose_h[i], verts_h[i, :len(g.verts)], frames_h[i, :len(g.verts)]) for i, g in enumerate(geoms)]


def create_block(target_block, target_face, geom, face, ox, oy):
    """AssemblyGym.create_block on the device.  target_block = None for the floor, else an object with
    ``verts_2d`` / ``geometry``.  Returns (pose[4], verts[nv,2], frames[nv,6]) float64 numpy."""
    return create_blocks([target_block], [target_face], [geom], [face], [ox], [oy])[0]


def pose_block(geom, pose4):
    """Block.__init__ on the device: world vertices and face frames of a shape posed by (x, z, cos, sin)."""
    L = abi.require_gpu()
    dev = device()
    sid = REGISTRY.id_of(geom)
    tab = REGISTRY.device_table()
    sh, p = upload(dev, np.array([sid], dtype=np.int32), np.array([list(pose4)], dtype=np.float64))
    verts = torch.empty((1, 6, 2), dtype=torch.float64, device=dev)
    frames = torch.empty((1, 6, 6), dtype=torch.float64, device=dev)
    abi.check(L.bridges_pose_block(tab, 1, _ptr(sh), _ptr(p), _ptr(verts), _stream()), "bridges_pose_block")
    abi.check(L.bridges_face_frames(tab, 1, _ptr(sh), _ptr(verts), _ptr(frames), _stream()), "bridges_face_frames")
    nv = len(geom.verts)
    verts_h, frames_h = download(verts, frames)
    return verts_h[0, :nv], frames_h[0, :nv]


def contains_points(block, points):
    """Shape.contains_2d (assembly_env.py:126-137): bool per (x, z) sample point, numpy in / numpy out."""
    L = abi.require_gpu()
    dev = device()
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(-1, 2))
    n = pts.shape[0]
    sid = REGISTRY.id_of(block.geometry)
    tab = REGISTRY.device_table()
    v = np.zeros((6, 2))
    v[:len(block.verts_2d)] = block.verts_2d
    tv, tp = upload(dev, v, pts)
    out = torch.zeros(n, dtype=torch.uint8, device=dev)
    abi.check(L.bridges_contains_points(tab, sid, _ptr(tv), n, _ptr(tp), _ptr(out), _stream()), "bridges_contains_points")
    return out.cpu().numpy().astype(bool)
