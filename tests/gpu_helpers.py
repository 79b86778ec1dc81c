"""Helpers shared by the GPU parity tests (not a test module)."""
import ctypes as C

import torch

from bridges_hip import abi


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def candidate_stability_unfused(env, chunk=8192):
    """The round-1 way of deciding is_action_stable_rbe for every valid candidate: gather a padded copy of the env's
    block list per candidate in torch, append the candidate, and let the stand-alone ``bridges_stability`` operator
    re-detect every interface from scratch.  Kept as an independent second path to cross-check the fused kernel at
    sizes the CPU oracle cannot reach.  Returns (rows, stable bool, error bool)."""
    L = abi.require_gpu()
    idx, row_env = env.valid_rows()
    n = idx.numel()
    K16 = abi.MAX_BLOCKS
    out = torch.zeros(n, dtype=torch.bool, device=env.device)
    errs = torch.zeros(n, dtype=torch.bool, device=env.device)
    if n == 0:
        return idx, out, errs
    ws_stride = abi.lp_ws_stride(K16)
    stab_ws = torch.empty((min(chunk, n), ws_stride), dtype=torch.float64, device=env.device)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for lo in range(0, n, chunk):
        ii, ee = idx[lo:lo + chunk], row_env[lo:lo + chunk]
        m = ii.numel()
        nb = env.buf["n_blocks"][ee].long()
        pose = torch.zeros((m, K16, 4), dtype=torch.float64, device=env.device)
        verts = torch.zeros((m, K16, 6, 2), dtype=torch.float64, device=env.device)
        shape = torch.zeros((m, K16), dtype=torch.int32, device=env.device)
        pose[:, :env.K] = env.buf["blk_pose"][ee]
        verts[:, :env.K] = env.buf["blk_verts"][ee]
        shape[:, :env.K] = env.buf["blk_shape"][ee]
        r = torch.arange(m, device=env.device)
        pose[r, nb] = env.buf["cand_pose"][ii]
        verts[r, nb] = env.buf["cand_verts"][ii]
        shape[r, nb] = env.buf["cand_desc"][ii, 2]
        nblocks = (nb + 1).to(torch.int32)
        fixed = torch.where(nb > 0, torch.ones_like(nb) << (nb - 1).clamp(min=0), torch.zeros_like(nb)).to(torch.int32)
        stable = torch.zeros(m, dtype=torch.uint8, device=env.device)
        info = torch.zeros((m, 8), dtype=torch.float64, device=env.device)
        fh, fd = env._create_args
        abi.check(L.bridges_stability(env.table.ptr, m, K16, _ptr(pose), _ptr(verts), _ptr(shape), _ptr(nblocks),
                                      _ptr(fixed), env.mu, env.density, fh, fd, _ptr(stable), _ptr(info),
                                      _ptr(stab_ws), ws_stride, stream), "bridges_stability")
        torch.cuda.synchronize()
        out[lo:lo + m] = stable.bool() & (info[:, 3] == 0)
        errs[lo:lo + m] = info[:, 3] != 0
    return idx, out, errs
