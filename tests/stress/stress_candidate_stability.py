"""Randomized parity of the fused candidate-stability kernel (bridges_env_candidate_stability) against the plain-C
oracle's is_action_stable_rbe (oracle/c: orc_candidate_stability), every valid candidate of every env, lock-step after
lock-step of a random-policy rollout.
    python tests/stress/stress_candidate_stability.py --envs 128 --locksteps 12 [--task tower4|hexbridge|mixed|...]"""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import numpy as np, torch
from oracle.c_env import CEnv
from oracle.env import OracleGym, bridge_setup, horizontal_bridge_setup
from bridges_hip.shapes import load_urdf
from bridges_hip.vec_env import VecAssemblyGym

ap = argparse.ArgumentParser()
ap.add_argument("--envs", type=int, default=128)
ap.add_argument("--locksteps", type=int, default=12)
ap.add_argument("--seed", type=int, default=23)
ap.add_argument("--task", default="tower4")
ap.add_argument("--density", type=float, default=1.0)
a = ap.parse_args()
TASKS = dict(tower4=(bridge_setup, dict(num_stories=4), ["trapezoid"], 15, 0.8),
             tower2=(bridge_setup, dict(num_stories=2), ["trapezoid"], 10, 0.8),
             hexbridge=(horizontal_bridge_setup, dict(num_obstacles=3, trapezoid=False, hexagon=True), ["hexagon"], 15, 0.8),
             mixed=(horizontal_bridge_setup, dict(num_obstacles=4, trapezoid=True, hexagon=True), ["trapezoid", "hexagon"], 12, 2.0),
             bridge_mu05=(horizontal_bridge_setup, dict(num_obstacles=5), ["trapezoid"], 15, 0.5))
fn, kw, names, max_steps, mu = TASKS[a.task]
setup = fn(**kw)
vec = VecAssemblyGym(a.envs, [load_urdf(f"shapes/{n}.urdf") for n in names], setup["obstacles"], setup["targets"],
                     max_steps=max_steps, seed=a.seed, mu=mu, density=a.density, f32_rasters=False)
gym = OracleGym(**setup, max_steps=max_steps, mu=mu, density=a.density)
cenvs = [CEnv(gym) for _ in range(a.envs)]
t0 = time.time(); checked = unstable = errors = mism = queued = 0
for it in range(a.locksteps):
    vec.candidate_stability_mask()
    off = vec.cand_offset.cpu().numpy(); mask = vec.cand_mask.cpu().numpy().astype(bool); cs = vec.cand_stable.cpu().numpy()
    queued += int(vec.cand_counters[0].item())
    for e, ce in enumerate(cenvs):
        ref = ce.candidate_stability()
        got = cs[off[e]:off[e] + len(ref)]
        m = mask[off[e]:off[e] + len(ref)]
        errors += int((got[m] == 2).sum())
        bad = np.flatnonzero((got == 1) != (ref == 1))
        checked += int(m.sum()); unstable += int((ref[m] == 0).sum())
        if len(bad):
            mism += len(bad)
            print("MISMATCH lock-step", it, "env", e, "candidates", bad[:8], "gpu", got[bad[:8]], "oracle", ref[bad[:8]])
            if mism > 20: sys.exit(1)
    vec.select_random()
    vec.step()
    for e, ce in enumerate(cenvs):
        ce.lockstep(a.seed, e)
    print(f"lock-step {it + 1}: {checked} candidates compared ({unstable} unstable), {mism} mismatches, {errors} lp errors, "
          f"{queued} large tableaux, {time.time() - t0:.0f} s", flush=True)
print(f"RESULT task={a.task} envs={a.envs} locksteps={a.locksteps} density={a.density}: {checked} candidate decisions compared "
      f"({unstable} unstable, {queued} through the large-tableau queue), {mism} mismatches, {errors} lp errors")
sys.exit(1 if (mism or errors) else 0)
