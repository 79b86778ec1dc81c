"""Multi-GPU plumbing of the vectorised DQN loop: one process per GPU, environments sharded by rank (no halo, no
data-path collective in the simulator), ONE all-gather per lock-step that replicates the new transition records
into every rank's replay ring (RCCL over xGMI on the GPU box: backend 'nccl'; 'gloo' for the CPU tests).

Records are fixed-size float64 rows (records.RECORD_WIDTH), padded per rank to the per-rank env count so a single
``all_gather_into_tensor`` suffices; the payload is tiny (E x 872 B per rank), i.e. latency-bound, not per-link
bandwidth-bound."""
import os

import torch
import torch.distributed as dist


def forced():
    """BRIDGES_FORCE_COLLECTIVE=1: a one-rank job still creates its process group and sends its records, module broadcasts
    and counters through the collectives (on a GPU: RCCL).  NCCL / RCCL refuse two ranks on one device, so a one-rank group
    is the only way to execute the multi-GPU branches on a one-GPU box (tests/test_gpu_one_rank_rccl.py)."""
    return os.environ.get("BRIDGES_FORCE_COLLECTIVE", "0") == "1"


def active():
    """True when records / parameters travel through a process group: several ranks, or one rank with forced()."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or forced())


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def init(backend=None, device=None):
    """Initialise torch.distributed from the torchrun environment; returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and not forced():
        return 0, 1
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:                                   # forced one-rank group started without torchrun
            os.environ.setdefault("MASTER_PORT", str(free_port()))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = dict(device_id=device) if (backend == "nccl" and device is not None) else {}
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def world_size():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def all_gather_records(rec, valid, n_valid=None):
    """rec [E, W] float64, valid [E] bool -> the valid records of every rank, rank-major ([N, W]).
    Every rank gets the same rows in the same order, so the replicated replay rings stay identical.
    n_valid = valid.sum() when the host already knows it: the single-rank selection then does not wait for the device
    (a boolean index reads its row count back)."""
    if not active():
        if n_valid is None:
            return rec[valid]
        return rec.index_select(0, torch.nonzero_static(valid, size=int(n_valid)).squeeze(1))
    world = dist.get_world_size()
    E, W = rec.shape
    payload = torch.cat([rec, valid.to(rec.dtype).unsqueeze(1)], dim=1).contiguous()     # validity travels in-band
    dev = rec.device
    if dist.get_backend() == "gloo" and payload.is_cuda:        # rehearsal mode: gloo moves host memory
        payload = payload.cpu()
    out = torch.empty((world * E, W + 1), dtype=rec.dtype, device=payload.device)
    dist.all_gather_into_tensor(out, payload)
    out = out.to(dev)
    keep = out[:, W] > 0.5
    return out[keep][:, :W]


def broadcast_module(module, src=0):
    """Re-synchronise replicated parameters (float atomics in backward can let replicas drift by ulps)."""
    if active():
        flat = getattr(module, "_flat_params", None)
        tensors = [flat.flat] if flat is not None else list(module.state_dict().values())
        for t in tensors:
            if dist.get_backend() == "gloo" and t.is_cuda:
                h = t.cpu()
                dist.broadcast(h, src)
                t.copy_(h)
            else:
                dist.broadcast(t, src)
