"""BASELINE.json configs[0] through the drop-in surface: the reference-style single-environment loop
(`successor_dqn.py --model=SuccessorMLP --tower_height=2 --num_episodes=50`, reference successor_dqn.py:697-781) on the
HIP operators.  Prints one JSON line: env-steps/s, seconds per episode, and the host synchronisations of one episode
(torch's sync-debug mode).  The reference's own recorded figures for this loop are 3-4 env-steps/s and 33-42 s per
episode."""
import argparse, json, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import torch
from robotoddler.training import successor_dqn as S

ap = argparse.ArgumentParser()
ap.add_argument("--episodes", type=int, default=50)
ap.add_argument("--warmup_episodes", type=int, default=5)
ap.add_argument("--model", default="SuccessorMLP")
ap.add_argument("--tower_height", type=int, default=2)
ap.add_argument("--loss", default="mse_q_values")
ap.add_argument("--count_syncs", action="store_true")
ap.add_argument("--profile", default="", help="write a cProfile listing (top 60 by cumulative time) of the timed run to this file")
a = ap.parse_args()
base = ["--model", a.model, "--tower_height", str(a.tower_height), "--loss_function", a.loss, "--seed", "0",
        "--evaluate_every", "1000000", "--learning_rate", "1e-4"]
S.main(base + ["--num_episodes", str(a.warmup_episodes)])          # warm-up: library load, first-use allocations
torch.cuda.synchronize()
prof = None
if a.profile:
    import cProfile
    prof = cProfile.Profile()
    prof.enable()
t0 = time.perf_counter()
hist = S.main(base + ["--num_episodes", str(a.episodes)])
torch.cuda.synchronize()
dt = time.perf_counter() - t0
if prof is not None:
    import io, pstats
    prof.disable()
    buf = io.StringIO()
    pstats.Stats(prof, stream=buf).sort_stats("cumulative").print_stats(60)
    pstats.Stats(prof, stream=buf).sort_stats("tottime").print_stats(40)
    open(a.profile, "w").write(buf.getvalue())
steps = sum(h["num_steps"] for h in hist)
out = dict(env_steps_per_s=steps / dt, s_per_episode=dt / len(hist), episodes=len(hist), env_steps=steps,
           config=f"successor_dqn.py --model={a.model} --tower_height={a.tower_height} --num_episodes={a.episodes} "
                  f"--loss_function={a.loss} (single env, 20 optimiser steps of batch 32 per episode)",
           reference_recorded="3-4 env-steps/s, 33-42 s per episode (BASELINE.md section 1: pyomo + IPOPT on CPU)")
if a.count_syncs:
    seen = []
    import collections, traceback
    where = collections.Counter()
    def hook(message, category, filename, lineno, file=None, line=None):
        if "synchroniz" in str(message):
            seen.append(1)
            # the innermost frame of this repo on the stack: which line of the drop-in surface made the host wait
            for fr in reversed(traceback.extract_stack()[:-1]):
                if ROOT in fr.filename and "tools/" not in fr.filename:
                    where[f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"] += 1
                    break
    warnings.showwarning = hook
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    h2 = S.main(base + ["--num_episodes", "3", "--num_training_steps", "0"])
    torch.cuda.set_sync_debug_mode("default")
    n2 = sum(h["num_steps"] for h in h2)
    out["host_syncs_per_env_step_rollout_only"] = len(seen) / max(n2, 1)
    out["host_syncs_by_site_per_env_step"] = {k: round(v / max(n2, 1), 2) for k, v in where.most_common(25)}
print(json.dumps(out))
