"""Lists the host synchronisations of one pipelined VecDQN lock-step (torch.cuda.set_sync_debug_mode): each one is a
point where the host stops queueing work until the GPU has caught up.  Usage: python tools/find_syncs.py [--model M]"""
import argparse
import os
import sys
import traceback
import warnings

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd"))
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="SuccessorMLP")
ap.add_argument("--envs", type=int, default=4096)
a = ap.parse_args()
from bridges_hip.shapes import load_urdf
from bridges_hip.vec_env import VecAssemblyGym
from robotoddler.training.successor_dqn import build_parser, make_nets
from robotoddler.training.vec_dqn import VecDQN

dev = torch.device("cuda:0")
args = vars(build_parser().parse_args(["--model", a.model]))
pol, tgt = make_nets(args, dev)
H = 0.8
env = VecAssemblyGym(a.envs, [load_urdf("shapes/trapezoid.urdf")], [(0.5, 0., i * H + H / 2) for i in range(4)],
                     [(0.5, 0, 4 * H + H / 2)], max_steps=15, seed=0, device=dev,
                     f32_rasters=VecDQN.acting_needs_f32_rasters(pol), candidate_snapshots=False)
agent = VecDQN(pol, tgt, torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True), env, 200000, 32, 0.95, 0.01,
               "mse_block_features")
for _ in range(5):
    agent.lockstep(25, defer_losses=True)
torch.cuda.synchronize()
seen = []


def hook(message, category, filename, lineno, file=None, line=None):
    if "synchroniz" in str(message):
        frames = [f for f in traceback.extract_stack() if "bridges" in f.filename or "robotoddler" in f.filename]
        seen.append(" <- ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in frames[-3:][::-1]))


warnings.showwarning = hook
warnings.simplefilter("always")
torch.cuda.set_sync_debug_mode("warn")
agent.lockstep(25, defer_losses=True)
torch.cuda.set_sync_debug_mode("default")
for s in seen:
    print(s)
print(len(seen), "synchronising calls in one lock-step")
