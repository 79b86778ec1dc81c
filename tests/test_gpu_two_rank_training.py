"""GPU: BASELINE.json configs[3]'s data path -- per-rank env shards, ONE all-gather of transition records per lock-step
into every rank's replica of the replay ring, replicated training -- with two ranks sharing the one card of the test box
(gloo moves the records through host memory; on a node the same calls run on RCCL).  Through the CLI entry point's
vectorised loop (successor_dqn.main -> run_vectorised), both losses of the config."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")

WORKER = r'''
import hashlib, json, os, sys
sys.path[:0] = [%(root)r, %(pkg)r]
import numpy as np, torch
from robotoddler.training import successor_dqn as S
from robotoddler.training.vec_dqn import run_vectorised
rank = int(os.environ["RANK"])
args = vars(S.build_parser().parse_args(
    ["--model", "SuccessorMLP", "--loss_function", "mse_q_values+mse_block_features", "--tower_height", "4",
     "--max_steps", "15", "--num_envs", "256", "--num_episodes", "1500", "--num_training_steps", "3", "--batch_size", "32",
     "--seed", "3", "--learning_rate", "1e-4", "--gamma", "0.95"]))
torch.cuda.set_device(0)
hist, agent = run_vectorised(args, torch.device("cuda", 0), return_agent=True)
torch.cuda.synchronize()
ring = agent.ring
order = (ring.head - ring.size + torch.arange(ring.size, device=ring.data.device)) %% ring.capacity
rec = ring.data[order].cpu().numpy()
w = agent.policy_net._flat_params.flat.detach().cpu().numpy()
wt = agent.target_net._flat_params.flat.detach().cpu().numpy()
losses = [h["avg_loss"] for h in hist if h["avg_loss"] is not None]
out = dict(rank=rank, locksteps=len(hist), ring_size=int(ring.size), ring_hash=hashlib.sha256(rec.tobytes()).hexdigest(),
           policy_hash=hashlib.sha256(w.tobytes()).hexdigest(), target_hash=hashlib.sha256(wt.tobytes()).hexdigest(),
           losses=losses, env_steps=int(agent.env_steps), episodes=int(agent.episodes_done),
           env_id_base=int(agent.env.env_id_base))
json.dump(out, open(os.path.join(%(tmp)r, "rank%%d.json" %% rank), "w"))
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
'''


def test_config4_two_ranks_gather_identical_rings_and_train_identical_replicas(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % dict(root=ROOT, pkg=PKG, tmp=str(tmp_path)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(MASTER_ADDR="127.0.0.1", BRIDGES_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:          # a free port, as bench.py's self-launch does
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    r = [json.load(open(tmp_path / f"rank{i}.json")) for i in range(2)]
    assert r[0]["locksteps"] == r[1]["locksteps"] >= 5            # both ranks stop on the same (gathered) episode count
    assert r[0]["ring_size"] == r[1]["ring_size"] > 256           # more records than one rank's envs produce per lock-step
    assert r[0]["ring_hash"] == r[1]["ring_hash"]                 # the gathered rings are bit-identical replicas
    assert r[0]["policy_hash"] == r[1]["policy_hash"]             # identical optimiser steps on identical batches
    assert r[0]["target_hash"] == r[1]["target_hash"]
    assert r[0]["episodes"] == r[1]["episodes"] >= 1500
    assert r[0]["env_id_base"] == 0 and r[1]["env_id_base"] == 256        # the env shards are disjoint
    assert r[0]["env_steps"] > 0 and r[1]["env_steps"] > 0
    for rr in r:
        assert len(rr["losses"]) >= 3 and all(l == l and 0.0 <= l < 1e6 for l in rr["losses"])    # finite, non-negative
    assert r[0]["losses"] == r[1]["losses"]
