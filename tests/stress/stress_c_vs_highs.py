"""CPU-only adjudication run: the plain-C oracle's two stability booleans of every env-step against HiGHS (numpy
oracle) on the same assemblies.  The C oracle shares its simplex rules with the device kernel (different arithmetic
details), so this measures how often those rules disagree with an exact LP solver.
    python tests/stress/stress_c_vs_highs.py --steps 2000 --envs 8 --procs 8"""
import argparse
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
TASKS = ["tower4", "hexbridge", "mixed", "tower2", "bridge_mu05"]


def make(task):
    from oracle.env import OracleGym, bridge_setup, horizontal_bridge_setup
    setup, max_steps, mu = dict(
        tower4=(bridge_setup(num_stories=4), 15, 0.8), tower2=(bridge_setup(num_stories=2), 10, 0.8),
        hexbridge=(horizontal_bridge_setup(num_obstacles=3, trapezoid=False, hexagon=True), 15, 0.8),
        mixed=(horizontal_bridge_setup(num_obstacles=4, trapezoid=True, hexagon=True), 12, 2.0),
        bridge_mu05=(horizontal_bridge_setup(num_obstacles=5), 15, 0.5))[task]
    return OracleGym(**setup, max_steps=max_steps, mu=mu)


def worker(args):
    task, env_id, n, seed = args
    from oracle.c_env import CEnv
    from oracle.geometry import Block
    from oracle.rbe import is_stable_rbe
    gym = make(task)
    ce = CEnv(gym)
    bad, steps, gap, prev = [], 0, 0, []
    for it in range(n):
        ncand, nvalid = C.c_int32(), C.c_int32()
        p = ce.L.orc_candidates(ce.h, C.byref(ncand), C.byref(nvalid))
        shapes_of = [p[i].sh for i in range(ncand.value)]
        o = ce.lockstep(seed, env_id)
        if not o.valid_step:
            prev = []
            continue
        steps += 1
        blocks = prev + [(shapes_of[o.action_index], (o.pose[0], o.pose[1]), (o.pose[2], o.pose[3]))]
        bl = [Block(gym.shapes[s], q, c) for s, q, c in blocks]
        for fixed, got in (({len(bl) - 1}, bool(o.stable_frozen)), (set(), bool(o.stable_unfrozen))):
            try:
                st, info = is_stable_rbe(bl, fixed, gym.mu, gym.density, gym.bounds, return_info=True)
            except RuntimeError as e:                       # HiGHS numerical failure: keep the case, count it as bad
                bad.append((task, env_id, it, sorted(fixed), got, repr(e)[:60], len(bl)))
                if os.environ.get("STRESS_DUMP"):
                    import pickle
                    with open(f"{os.environ['STRESS_DUMP']}/fail_{task}_{env_id}_{it}_{len(fixed)}.pkl", "wb") as fh:
                        pickle.dump((blocks, sorted(fixed), gym.mu), fh)
                continue
            if info["v"] is not None and 1e-6 < info["v"] < 1e-4:
                gap += 1
            if st != got:
                bad.append((task, env_id, it, sorted(fixed), got, info["v"], len(bl)))
        prev = [] if o.done else blocks
    return steps, bad, gap


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=2000, help="lock-steps per (task, env)")
    ap.add_argument("--envs", type=int, default=8)
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--seed", type=int, default=101)
    ap.add_argument("--tasks", default=",".join(TASKS))
    a = ap.parse_args()
    from oracle import c_env
    c_env.lib()
    jobs = [(t, e, a.steps, a.seed) for t in a.tasks.split(",") for e in range(a.envs)]
    t0 = time.time()
    with mp.get_context("fork").Pool(a.procs) as pool:
        res = pool.map(worker, jobs)
    steps = sum(r[0] for r in res)
    bad = [b for r in res for b in r[1]]
    gap = sum(r[2] for r in res)
    print(f"{steps} env-steps ({2 * steps} stability decisions) on {a.tasks}: {len(bad)} disagreements with HiGHS, "
          f"{gap} LPs with v* inside (1e-6, 1e-4), {time.time() - t0:.0f} s")
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
