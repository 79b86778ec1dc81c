"""GPU: the RCCL branches of the multi-GPU path executed with a ONE-rank process group (BRIDGES_FORCE_COLLECTIVE=1).

RCCL refuses two ranks on one device and the test box has one card, so a one-rank `nccl` group is the only rehearsal
of `dist.init_process_group("nccl", device_id=...)`, `all_gather_into_tensor` of float64 device records and the device
branch of `broadcast_module` this pool allows.  Every job here is a fresh child process started under
`python -m torch.distributed.run --nproc-per-node=1` before anything touches the GPU.  N > 1 on hardware stays
unmeasured by these tests."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")


def _port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "BRIDGES_DIST_BACKEND", "BENCH_DIST_BACKEND")}
    env.update(MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", **extra)
    return env


def _torchrun(script_and_args, env, timeout=900):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_port()), *script_and_args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)


def test_config4_bench_leg_on_a_one_rank_rccl_group():
    """bench.py's configs[3] leg (per-rank env shard, one all_gather_into_tensor of records per lock-step, replicated
    training) with backend nccl: init_process_group("nccl", device_id=...), the RCCL all-gather of device float64 rows,
    the all_gather / all_reduce of the hashes and times on device tensors."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--config4-leg", "--gpus", "1", "--envs", "512"],
                    _env(BRIDGES_FORCE_COLLECTIVE="1", BENCH_TRAIN_LOCKSTEPS="4", BENCH_TRAIN_WARMUP="4"))
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    leg = json.loads(lines[0])
    assert leg["dist_backend"] == "nccl" and leg["ranks_seen"] == 1
    assert leg["ring_hash_equal"] is True and leg["policy_hash_equal"] is True and leg["last_losses_finite"] is True
    assert leg["allgather_rows_received"] == 512 and leg["allgather_ms_per_lockstep"] > 0
    assert leg["ring_records"] > 512 and leg["value"] > 1e3


def test_headline_bench_and_its_child_leg_on_a_one_rank_rccl_group():
    """`bench.py --gpus 1` under torchrun with the forced process group: the N > 1 flow end to end on one GPU -- nccl init with
    device_id, the barriers and the MAX / SUM all-reduces of the timing contract on device tensors, destroy_process_group, then
    the configs[3] leg started by rank 0 as a child torchrun job with its own nccl group -- and ONE JSON line at the end."""
    out = _torchrun([os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "10", "--warmup", "3", "--envs", "512", "--no-other-modes",
                     "--no-cpu-baseline"],
                    _env(BRIDGES_FORCE_COLLECTIVE="1", BENCH_TRAIN_LOCKSTEPS="3", BENCH_TRAIN_WARMUP="3"))
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["ranks_seen"] == 1 and j["config"]["dist_backend"] == "nccl" and j["scaling"] == "weak"
    assert j["value"] > 1e5 and len(j["config"]["per_seed"]) == 3
    leg = j["other_modes"]["train_config4"]
    assert "error" not in leg, leg
    assert leg["dist_backend"] == "nccl" and leg["ranks_seen"] == 1 and leg["ring_hash_equal"] is True and leg["policy_hash_equal"] is True
    assert leg["allgather_rows_received"] == 512 and leg["value"] > 1e3


WORKER = r'''
import hashlib, json, os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np, torch
import torch.distributed as dist
from robotoddler.training import distributed as D
from robotoddler.training import successor_dqn as S
from robotoddler.training.vec_dqn import run_vectorised
args = vars(S.build_parser().parse_args(
    ["--model", "SuccessorMLP", "--loss_function", "mse_q_values+mse_block_features", "--tower_height", "4",
     "--max_steps", "15", "--num_envs", "256", "--num_episodes", "900", "--num_training_steps", "3", "--batch_size", "32",
     "--seed", "3", "--learning_rate", "1e-4", "--gamma", "0.95"]))
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
hist, agent = run_vectorised(args, dev, return_agent=True)
torch.cuda.synchronize()
ring = agent.ring
order = (ring.head - ring.size + torch.arange(ring.size, device=ring.data.device)) %% ring.capacity
rec = ring.data[order].cpu().numpy()
w = agent.policy_net._flat_params.flat.detach().cpu().numpy()
out = dict(locksteps=len(hist), ring_size=int(ring.size), ring_hash=hashlib.sha256(rec.tobytes()).hexdigest(),
           policy_hash=hashlib.sha256(w.tobytes()).hexdigest(), episodes=int(agent.episodes_done), env_steps=int(agent.env_steps),
           losses=[h["avg_loss"] for h in hist if h["avg_loss"] is not None],
           active=bool(D.active()), backend=(dist.get_backend() if dist.is_initialized() else None),
           world=(dist.get_world_size() if dist.is_initialized() else None))
if D.active():
    # the collective alone: float64 device rows through RCCL, against the single-rank selection rec[valid]
    g = torch.Generator(device=dev).manual_seed(11)
    r = torch.rand((300, ring.data.shape[1]), generator=g, device=dev, dtype=torch.float64)
    valid = torch.rand(300, generator=g, device=dev) < 0.6
    got = D.all_gather_records(r, valid)
    out["gather_equal"] = bool(torch.equal(got, r[valid])) and got.is_cuda and got.dtype == torch.float64
    none = D.all_gather_records(r, torch.zeros(300, dtype=torch.bool, device=dev))
    out["gather_empty_rows"] = int(none.shape[0])
    # broadcast_module: the flat-parameter branch (SuccessorMLP) and the per-tensor branch (a plain module), on device tensors
    before = agent.policy_net._flat_params.flat.detach().clone()
    D.broadcast_module(agent.policy_net)
    lin = torch.nn.Linear(5, 3).to(dev)
    wl = lin.weight.detach().clone()
    D.broadcast_module(lin)
    torch.cuda.synchronize()
    out["broadcast_keeps_weights"] = bool(torch.equal(before, agent.policy_net._flat_params.flat)) and bool(torch.equal(wl, lin.weight))
    dist.barrier()
    dist.destroy_process_group()
json.dump(out, open(%(out)r, "w"))
'''


def test_vectorised_loop_through_rccl_equals_the_plain_single_rank_loop(tmp_path):
    """run_vectorised for ~10 lock-steps twice with the same seed: once plain (no process group: rec[valid] goes straight
    into the ring), once with a one-rank nccl group (every lock-step's records pass through all_gather_into_tensor on
    the device, episode counts come from the gathered rows, parameters are broadcast).  Rings and weights must agree bit
    for bit."""
    res = {}
    for name, extra in (("plain", {}), ("rccl", dict(BRIDGES_FORCE_COLLECTIVE="1"))):
        script = tmp_path / f"worker_{name}.py"
        outp = tmp_path / f"{name}.json"
        script.write_text(WORKER % dict(root=ROOT, pkg=PKG, out=str(outp)))
        out = _torchrun([str(script)], _env(**extra))
        assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
        res[name] = json.load(open(outp))
    p, r = res["plain"], res["rccl"]
    assert p["active"] is False and p["backend"] is None
    assert r["active"] is True and r["backend"] == "nccl" and r["world"] == 1
    assert r["gather_equal"] is True and r["gather_empty_rows"] == 0 and r["broadcast_keeps_weights"] is True
    assert p["locksteps"] == r["locksteps"] >= 5 and p["episodes"] == r["episodes"] >= 900
    assert p["env_steps"] == r["env_steps"] > 0
    assert p["ring_size"] == r["ring_size"] > 256
    assert p["ring_hash"] == r["ring_hash"]                      # the gathered ring is the ring of the non-collective path
    assert p["policy_hash"] == r["policy_hash"]
    assert p["losses"] == r["losses"] and len(p["losses"]) >= 3
