"""Does the vectorised loop learn?  Trains a Q-network (--model SuccessorMLP | ConvNet | UNet) on tower_height=2 for a fixed
number of lock-steps and prints the mean sparse reward, linear reward and loss per block of lock-steps (evaluation = the running
epsilon-greedy rollouts).
    python tools/learning_curve.py --locksteps 1500 --envs 1024 [--model ConvNet --loss mse_q_values]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import numpy as np
import torch
from robotoddler.training.successor_dqn import build_parser, make_nets
from robotoddler.training.vec_dqn import VecDQN
from robotoddler.training import records as R
from bridges_hip.shapes import load_urdf
from bridges_hip.vec_env import VecAssemblyGym

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=1024)
ap.add_argument("--locksteps", type=int, default=1500)
ap.add_argument("--tower", type=int, default=2)
ap.add_argument("--max_steps", type=int, default=10)
ap.add_argument("--train_steps", type=int, default=10)
ap.add_argument("--loss", default="mse_q_values+mse_block_features")
ap.add_argument("--lr", type=float, default=1e-4)
ap.add_argument("--block", type=int, default=100)
ap.add_argument("--model", default="SuccessorMLP")
a = ap.parse_args()
dev = torch.device("cuda:0")
args = vars(build_parser().parse_args(["--model", a.model, "--loss_function", a.loss]))
H = 0.8
torch.manual_seed(0)
pol, tgt = make_nets(args, dev)
env = VecAssemblyGym(a.envs, [load_urdf("shapes/trapezoid.urdf")], [(0.5, 0., i * H + H / 2) for i in range(a.tower)],
                     [(0.5, 0, a.tower * H + H / 2)], max_steps=a.max_steps, seed=0, device=dev,
                     f32_rasters=VecDQN.acting_needs_f32_rasters(pol))
agent = VecDQN(pol, tgt, torch.optim.Adam(pol.parameters(), lr=a.lr, fused=True), env, 200000, 32, 0.95, 0.01, a.loss,
               eps_decay=0.997)
t0 = time.time()
acc = dict(steps=0, reward=0.0, lin=0.0, done=0, solved=0, loss=[])
for it in range(1, a.locksteps + 1):
    losses, rec = agent.lockstep(a.train_steps)
    acc["steps"] += rec.shape[0]
    acc["reward"] += float(rec[:, R.O_REWARD].sum()); acc["lin"] += float(rec[:, R.O_LIN].sum())
    d = rec[:, R.O_DONE] > 0.5
    acc["done"] += int(d.sum()); acc["solved"] += int((d & (rec[:, R.O_REWARD] > 0.5)).sum())
    acc["loss"] += losses
    if it % a.block == 0:
        print(json.dumps(dict(lockstep=it, seconds=round(time.time() - t0, 1), epsilon=round(agent.epsilon, 3),
                              mean_reward=round(acc["reward"] / max(acc["steps"], 1), 4),
                              mean_lin_reward=round(acc["lin"] / max(acc["steps"], 1), 4),
                              episodes=acc["done"], solved_fraction=round(acc["solved"] / max(acc["done"], 1), 4),
                              mean_loss=round(float(np.mean(acc["loss"])), 5) if acc["loss"] else None)), flush=True)
        acc = dict(steps=0, reward=0.0, lin=0.0, done=0, solved=0, loss=[])
