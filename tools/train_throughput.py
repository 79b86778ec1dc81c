"""Training-in-the-loop throughput of the vectorised successor-DQN (reported separately from the simulator bench)."""
import argparse, json, os, sys, time
if "--miopen_search" not in sys.argv:
    os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import torch
from robotoddler.training.successor_dqn import build_parser, make_nets
from robotoddler.training.vec_dqn import VecDQN
from bridges_hip.shapes import load_urdf
from bridges_hip.vec_env import VecAssemblyGym

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=4096)
ap.add_argument("--tower", type=int, default=4)
ap.add_argument("--max_steps", type=int, default=15)
ap.add_argument("--model", default="SuccessorMLP")
ap.add_argument("--loss", default="mse_block_features")
ap.add_argument("--locksteps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=8)
ap.add_argument("--train_steps", type=int, default=25)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--no_fused_adam", action="store_true")
ap.add_argument("--no_dedup", action="store_true",
                help="A/B: every env its own candidate rows and every row fed, also when envs are in the same state / rows have identical inputs")
ap.add_argument("--miopen_search", action="store_true", help="let MIOpen benchmark its algorithms (one shape per run)")
ap.add_argument("--channels_last", action="store_true")
ap.add_argument("--shapes", default="trapezoid", choices=["trapezoid", "hexagon", "both"])
ap.add_argument("--bridge_length", type=int, default=0, help="> 0: horizontal_bridge_setup(num_obstacles=N) instead of the tower")
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.miopen_search:
    torch.backends.cudnn.benchmark = True
args = vars(build_parser().parse_args(["--model", a.model, "--loss_function", a.loss, "--learning_rate", "1e-4"]))
H = 0.8
torch.manual_seed(0)
pol, tgt = make_nets(args, dev)
if a.channels_last:
    pol, tgt = pol.to(memory_format=torch.channels_last), tgt.to(memory_format=torch.channels_last)
names = dict(trapezoid=["trapezoid"], hexagon=["hexagon"], both=["trapezoid", "hexagon"])[a.shapes]
if a.bridge_length:
    sq, nn = 0.6, a.bridge_length
    obstacles, targets = [(i * sq, 0.0, sq / 2) for i in range(1, nn + 1)], [(nn * sq + 2.5 * sq, 0.0, sq / 2)]
else:
    obstacles, targets = [(0.5, 0., i * H + H / 2) for i in range(a.tower)], [(0.5, 0, a.tower * H + H / 2)]
env = VecAssemblyGym(a.envs, [load_urdf(f"shapes/{n}.urdf") for n in names], obstacles, targets, max_steps=a.max_steps, seed=0,
                     device=dev, f32_rasters=VecDQN.acting_needs_f32_rasters(pol), candidate_snapshots=False)
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=not a.no_fused_adam)
agent = VecDQN(pol, tgt, opt, env, 200000, a.batch, 0.95, 0.01, a.loss)
VecDQN.TRACK_ROWS = True
if a.no_dedup:
    VecDQN.DEDUP_ROWS = VecDQN.DEDUP_STATES = False
for i in range(a.warmup):
    agent.lockstep(a.train_steps)
    print("warm-up lock-step", i, "done", flush=True)
# 1) the loop as run_vectorised runs it: nothing between lock-steps waits for the optimiser steps (deferred loss readback)
agent._rows_seen_dev, agent.rows_fed = None, 0
torch.cuda.synchronize(); s0 = agent.env_steps; t0 = time.perf_counter()
pending, per_step, tp = None, [], t0
for _ in range(a.locksteps):
    deferred, _rec = agent.lockstep(a.train_steps, defer_losses=True)
    if pending is not None:
        pending.get()
    pending = deferred
    tn = time.perf_counter(); per_step.append(tn - tp); tp = tn
losses = pending.get()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
steps_done = agent.env_steps - s0
rows_seen, rows_fed = getattr(agent, "rows_seen", 0), getattr(agent, "rows_fed", 0)
per_step.sort()
median = per_step[len(per_step) // 2]
# 2) the same lock-steps with a device synchronisation between the phases, for the per-phase times only
t_targets = [0.0]
_orig_targets = agent._targets
def _timed_targets(rec):
    torch.cuda.synchronize(); ta = time.perf_counter()
    out = _orig_targets(rec)
    torch.cuda.synchronize(); t_targets[0] += time.perf_counter() - ta
    return out
agent._targets = _timed_targets
t_act = t_train = 0.0
n_phase = max(2, a.locksteps // 2)
for _ in range(n_phase):
    torch.cuda.synchronize(); t1 = time.perf_counter()
    rec, valid = agent.act()
    agent.env_steps += int(valid.sum().item())
    agent.ring.push(rec[valid])
    torch.cuda.synchronize(); t2 = time.perf_counter()
    agent.train_steps(a.train_steps)
    agent.update_target()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    t_act += t2 - t1; t_train += t3 - t2
print(json.dumps(dict(config=vars(a), env_steps_per_s=steps_done / dt,
                      env_steps_per_s_at_median_lockstep=steps_done / a.locksteps / median, ms_median_lockstep=median * 1e3,
                      ms_per_lockstep=dt / a.locksteps * 1e3,
                      rows_per_lockstep=rows_seen / a.locksteps, rows_fed_fraction=(rows_fed / rows_seen) if rows_seen else None,
                      ms_act=t_act / n_phase * 1e3, ms_targets=t_targets[0] / n_phase * 1e3, ms_train=t_train / n_phase * 1e3,
                      ms_per_train_step=(t_train - t_targets[0]) / n_phase / a.train_steps * 1e3, last_loss=losses[-1] if losses else None,
                      note="env_steps_per_s: pipelined loop (VecDQN.lockstep, deferred loss readback); ms_act / ms_targets / "
                           "ms_train: separate pass with a device synchronisation between the phases (they overlap in the loop)")))
