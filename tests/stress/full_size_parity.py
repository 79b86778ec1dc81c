"""Oracle parity at BASELINE.json's full sizes: the HIP lock-step environment against the plain-C oracle (oracle/c) for
every env and every lock-step -- selected action, both stability booleans, termination / truncation, reward, linear
reward, targets reached, block count, candidate / valid counts and the state's bit raster; with --candidates also
is_action_stable_rbe of EVERY valid candidate of every env after every `--candidates`-th lock-step (the fused
candidate-stability kernel against the oracle's orc_candidate_stability).

The policy's draws are keyed by (seed, global env id, draw counter), so the oracle's trajectories do not depend on the
GPU's: the oracle shards run FIRST in a fork pool (one env at a time per core, before this process touches the GPU),
then the HIP path runs the same lock-steps and the recorded arrays are compared.

    python tests/stress/full_size_parity.py --config 3 [--locksteps 25] [--workers 16]
        --config 3: BASELINE configs[2]  4096 envs, tower_height=4, max_steps=15, trapezoid
        --config 2: BASELINE configs[1]  1024 envs, tower_height=2, max_steps=10, trapezoid  (the simulator side)
        --config 5: BASELINE configs[4]  4096 envs, hexagon, horizontal bridge of 3, max_steps=15 (one GPU's share)
        --config 6: 1024 envs, trapezoid + hexagon, horizontal bridge of 4, mu = 2, max_steps=12 (marginal LP optima)
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import numpy as np

CONFIGS = {2: dict(envs=1024, setup="bridge", kw=dict(num_stories=2), names=["trapezoid"], max_steps=10),
           3: dict(envs=4096, setup="bridge", kw=dict(num_stories=4), names=["trapezoid"], max_steps=15),
           5: dict(envs=4096, setup="hbridge", kw=dict(num_obstacles=3, trapezoid=False, hexagon=True), names=["hexagon"], max_steps=15),
           # not a BASELINE config: two shapes, mu = 2 -- the task on which a continued tableau once reported "stable" at an optimum
           # of 1.41e-5 (threshold 1e-5; --seed 99, lock-step 88, env 797: rbe_device.h LP_MARGIN_LO)
           6: dict(envs=1024, setup="hbridge", kw=dict(num_obstacles=4, trapezoid=True, hexagon=True), names=["trapezoid", "hexagon"],
                   max_steps=12, mu=2.0),
           # further stress tasks (tests/stress only): low / high friction, long episodes that fill the 16 block slots
           7: dict(envs=1024, setup="hbridge", kw=dict(num_obstacles=5), names=["trapezoid"], max_steps=15, mu=0.5),
           8: dict(envs=1024, setup="hbridge", kw=dict(num_obstacles=6, trapezoid=True, hexagon=True), names=["trapezoid", "hexagon"],
                   max_steps=15, mu=1.0),
           9: dict(envs=1024, setup="bridge", kw=dict(num_stories=6, hexagon=True), names=["trapezoid", "hexagon"], max_steps=15, mu=3.0)}
FIELDS = ("valid_step", "no_actions", "action_index", "stable_frozen", "stable_unfrozen", "terminated", "truncated", "done",
          "n_blocks", "n_reached")


def _setup(cfg):
    from oracle.env import bridge_setup, horizontal_bridge_setup
    return dict(bridge=bridge_setup, hbridge=horizontal_bridge_setup)[cfg["setup"]](**cfg["kw"])


def oracle_shard(job):
    """Lock-steps of envs e0..e1 on the C oracle: dict of [L, n] arrays (+ state bits [L, n, 64])."""
    import ctypes as C
    cfg, seed, e0, e1, L, cand_every = job
    from oracle.c_env import CEnv, IMG
    from oracle.env import OracleGym
    gym = OracleGym(**_setup(cfg), max_steps=cfg["max_steps"], mu=cfg.get("mu", 0.8), density=cfg.get("density", 1.0))
    n = e1 - e0
    out = {k: np.zeros((L, n), dtype=np.int32) for k in FIELDS + ("n_cand", "n_valid")}
    out["reward"] = np.zeros((L, n))
    out["lin_reward"] = np.zeros((L, n))
    out["state_bits"] = np.zeros((L, n, IMG), dtype=np.uint64)
    a_max = CEnv(gym).cfg.a_max
    n_cs = len(range(cand_every - 1, L, cand_every)) if cand_every else 0
    out["cand_stable"] = np.zeros((n_cs, n, a_max), dtype=np.uint8)       # 1 stable, 0 unstable or masked out
    envs = [CEnv(gym) for _ in range(n)]
    nc, nv = C.c_int32(), C.c_int32()
    for i, ce in enumerate(envs):
        for it in range(L):
            o = ce.lockstep(seed, e0 + i)
            for k in FIELDS:
                out[k][it, i] = getattr(o, k)
            out["reward"][it, i], out["lin_reward"][it, i] = o.reward, o.lin_reward
            ce.L.orc_candidates(ce.h, C.byref(nc), C.byref(nv))
            out["n_cand"][it, i], out["n_valid"][it, i] = nc.value, nv.value
            out["state_bits"][it, i] = np.ctypeslib.as_array(ce.L.orc_state_bits(ce.h), shape=(IMG,))
            if cand_every and it % cand_every == cand_every - 1:
                cs = ce.candidate_stability()
                out["cand_stable"][it // cand_every, i, :cs.size] = cs
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--envs", type=int, default=0, help="override the config's env count")
    ap.add_argument("--locksteps", type=int, default=25)
    ap.add_argument("--seed", type=int, default=41)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--groups", type=int, default=2, help="env groups on their own HIP streams, as bench.py runs them")
    ap.add_argument("--sparse", action="store_true", help="sparse row-group update of the f32 rasters (only changed rows are rewritten): "
                    "every candidate's and every state's f32 image is compared with the expansion of its bit raster after EVERY lock-step")
    ap.add_argument("--density", type=float, default=0.0, help="override the block density (every force tolerance scales with it)")
    ap.add_argument("--mu", type=float, default=0.0, help="override the friction coefficient")
    ap.add_argument("--candidates", type=int, default=0, metavar="N",
                    help="> 0: also compare is_action_stable_rbe of every valid candidate after every N-th lock-step")
    a = ap.parse_args()
    cfg = dict(CONFIGS[a.config])
    if a.density:
        cfg["density"] = a.density
    if a.mu:
        cfg["mu"] = a.mu
    E = a.envs or cfg["envs"]
    workers = a.workers or max(1, min(len(os.sched_getaffinity(0)), 16))
    from oracle import c_env
    c_env.lib()                                                  # build / load once before forking
    t0 = time.time()
    per = -(-E // (workers * 4))                                 # 4 jobs per worker: even out the episode-length lottery
    jobs = [(cfg, a.seed, e0, min(E, e0 + per), a.locksteps, a.candidates) for e0 in range(0, E, per)]
    with mp.get_context("fork").Pool(workers) as pool:
        parts = pool.map(oracle_shard, jobs)
    ora = {k: np.concatenate([p[k] for p in parts], axis=1) for k in parts[0]}
    ora_cs = ora.pop("cand_stable")
    t_oracle = time.time() - t0

    import torch                                                  # only now: the pool is gone, nothing forked holds the GPU
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGymGroups
    setup = _setup(cfg)
    t1 = time.time()
    vec = VecAssemblyGymGroups(E, [load_urdf(f"shapes/{n}.urdf") for n in cfg["names"]], setup["obstacles"], setup["targets"],
                               groups=a.groups, max_steps=cfg["max_steps"], seed=a.seed, f32_rasters=True, mu=cfg.get("mu", 0.8), density=cfg.get("density", 1.0),
                               sparse_raster_update=a.sparse,
                               candidate_snapshots=bool(a.candidates))
    mism, steps, lp_err, overflow = 0, 0, 0, 0
    f32_every = True
    cs_decisions = cs_mism = cs_err = 0
    cat = lambda name: torch.cat([getattr(g, name) for g in vec.envs]).cpu().numpy()
    for it in range(a.locksteps):
        for g, st in zip(vec.envs, vec.streams):                 # select, read the selection back, then step (per group stream)
            with torch.cuda.stream(st):
                g.select_random()
        vec.sync()
        sel = cat("sel_index")
        for g, st in zip(vec.envs, vec.streams):
            with torch.cuda.stream(st):
                g.step()
        vec.sync()
        fl, rew, lin, nre = cat("step_flags"), cat("reward"), cat("lin_reward"), cat("n_reached")
        nb, ncand, nval = cat("n_blocks"), cat("n_cand"), cat("n_valid")
        sb = cat("state_bits").view(np.uint64)
        o = {k: v[it] for k, v in ora.items()}
        v = o["valid_step"].astype(bool)
        ok = (fl[:, 0].astype(bool) == v) & (fl[:, 6].astype(bool) == o["no_actions"].astype(bool))
        step_ok = ((sel == o["action_index"]) & (fl[:, 1] == o["stable_frozen"]) & (fl[:, 2] == o["stable_unfrozen"])
                   & (fl[:, 3] == o["terminated"]) & (fl[:, 4] == o["truncated"]) & (fl[:, 5] == o["done"])
                   & (rew == o["reward"]) & (nre == o["n_reached"]) & ((fl[:, 7] & 3) == 0)
                   & np.isclose(lin, o["lin_reward"], rtol=1e-5, atol=1e-7))
        ok &= np.where(v, step_ok, True)
        # after the lock-step (auto-reset included) both sides hold the same state and the same candidate set
        nb_after = np.where(v & (o["done"] == 0), o["n_blocks"], 0)          # the oracle reports the count before its auto-reset
        ok &= (nb == nb_after) & (ncand == o["n_cand"]) & (nval == o["n_valid"]) & (sb == o["state_bits"]).all(axis=1)
        if a.candidates and it % a.candidates == a.candidates - 1:
            # is_action_stable_rbe of every valid candidate of the states just reached, group by group
            e_base = 0
            for g, st in zip(vec.envs, vec.streams):
                with torch.cuda.stream(st):
                    g.candidate_stability_mask()
                st.synchronize()
                off, nc = g.cand_offset.cpu().numpy(), g.n_cand.cpu().numpy()
                got, msk = g.cand_stable.cpu().numpy(), g.cand_mask.cpu().numpy()
                want = ora_cs[it // a.candidates, e_base:e_base + g.E]
                for e in range(g.E):
                    sl = slice(off[e], off[e] + nc[e])
                    valid = msk[sl] != 0
                    cs_decisions += int(valid.sum())
                    cs_err += int((got[sl][valid] == 2).sum())
                    bad_c = int((got[sl] != want[e, :nc[e]]).sum())
                    if bad_c and cs_mism < 5:
                        print("CANDIDATE MISMATCH lock-step", it, "env", e_base + e, "gpu", got[sl].tolist(), "oracle", want[e, :nc[e]].tolist())
                    cs_mism += bad_c
                e_base += g.E
        steps += int(v.sum())
        lp_err += int(((fl[:, 7] & 1) != 0).sum())
        overflow += int(((fl[:, 7] & 2) != 0).sum())
        if a.sparse:                                             # the incremental f32 images never hold a stale row
            from bridges_hip import ops
            for g in vec.envs:
                n = g.total_candidates()
                f32_every &= bool(torch.equal(g.cand_raster[:n], ops.bits_to_f32(g.cand_bits[:n])))
                f32_every &= bool(torch.equal(g.state_raster, ops.bits_to_f32(g.state_bits)))
        bad = np.flatnonzero(~ok)
        mism += bad.size
        for e in bad[:5]:
            print("MISMATCH lock-step", it, "env", e, "flags", fl[e], "sel", sel[e], "oracle", {k: o[k][e] for k in FIELDS + ("n_cand", "n_valid", "reward", "lin_reward")},
                  "gpu", dict(reward=rew[e], lin=lin[e], n_blocks=nb[e], n_cand=ncand[e], n_valid=nval[e]))
    # the f32 images of the final candidate sets are the expansion of their bit rasters (spot check of the 16 KiB-per-candidate path)
    f32_ok = f32_every
    for g in vec.envs:
        tot = g.total_candidates()
        n = min(tot, 4096)
        from bridges_hip import ops
        f32_ok &= bool(torch.equal(g.cand_raster[:n], ops.bits_to_f32(g.cand_bits[:n])))
        f32_ok &= bool(torch.equal(g.state_raster, ops.bits_to_f32(g.state_bits)))
    stats = vec.read_stats()
    print(f"RESULT config={a.config} envs={E} locksteps={a.locksteps} groups={a.groups}: {steps} env-steps ({2 * steps} stability decisions) "
          f"compared, {mism} mismatches, lp_errors={lp_err} contact_overflows={overflow} cand_overflow={stats.get('cand_overflow', 0)} "
          f"f32_equals_bits={f32_ok}; candidate_decisions={cs_decisions} candidate_mismatches={cs_mism} candidate_errors={cs_err}; "
          f"oracle {t_oracle:.1f} s on {workers} workers, gpu + compare {time.time() - t1:.1f} s")
    sys.exit(0 if (mism == 0 and f32_ok and steps > 0 and cs_mism == 0 and cs_err == 0) else 1)


if __name__ == "__main__":
    main()
