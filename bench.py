#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the vectorised assembly_gym hot path (BASELINE.json metric).

One "step" of this bench = one lock-step of the hot path over one batch of E synthetic
environments: uniform-random policy draw -> place the block -> contact interfaces -> stability
{last frozen, none frozen} -> reward/termination/auto-reset -> enumerate + place the A raw
candidates -> (A+1) 64x64 f32 rasters -> bounds/overlap mask -> linear reward  (SURVEY.md §8d).
`value` counts real environment steps (reset-only lock-steps of an env are not counted).

  python bench.py [--gpus N --steps K --warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Envs are independent: each rank owns E envs (weak scaling), there is no data-path collective;
the only communication is the barrier / max-reduce of the timing contract.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(sum_cand, sum_blocks, n_envsteps_units, V=4):
    """bytes one lock-step must move for `n_envsteps_units` environments holding
    sum_blocks blocks and sum_cand raw candidates in total (f32 rasters as the reference surfaces them)."""
    A, k, E = float(sum_cand), float(sum_blocks), float(n_envsteps_units)
    return (4 * 64 * 64 * (A + E)          # action + state rasters written
            + 2 * 4 * 64 * 64 * E          # state + obstacle rasters read for the overlap test
            + 8 * 2 * V * (k + A)          # vertex reads, f64
            + 24 * (k + A)                 # poses
            + 16 * A + A + 4 * A           # candidate descriptors, mask, lin_reward
            + 100 * E)                     # stability I/O


# ------------------------------------------------------------------------- CPU baseline (oracle)
def _cpu_worker(args):
    seed, env_id, seconds, tower_height, max_steps = args
    from oracle.c_env import CEnv
    from oracle.env import OracleGym, bridge_setup
    ce = CEnv(OracleGym(**bridge_setup(num_stories=tower_height), max_steps=max_steps))
    ce.enable_f32()                                  # same unit of work: f32 rasters for every raw candidate
    ce.run(seed, env_id, 200)                        # warm-up
    n, chunk = 0, 2000
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        n += ce.run(seed, env_id, chunk)
    return n, time.perf_counter() - t0


def cpu_baseline(tower_height, max_steps, seconds=10.0):
    """The plain-C restatement of the path (oracle/c, tested bit for bit against the numpy + HiGHS oracle), one
    environment per process on every host core."""
    import multiprocessing as mp
    from oracle import c_env
    c_env.lib()                                      # build once before forking
    # a one-GPU box of this pool gives the job a 16-core share whatever the affinity mask says
    seconds = float(os.environ.get("BENCH_CPU_SECONDS", seconds))          # tests shorten the sample
    cores = int(os.environ.get("BENCH_CPU_WORKERS", max(1, min(len(os.sched_getaffinity(0)), 16))))
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(0, i, seconds, tower_height, max_steps) for i in range(cores)])
    steps = sum(r[0] for r in res)
    wall = max(r[1] for r in res)
    return dict(value=steps / wall, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} processes x {seconds:.0f} s of tower_height={tower_height} random-policy lock-steps "
                       f"({steps} env-steps) with oracle/c/oracle_env.c (scalar C, -O2, float64, own simplex, "
                       f"bit + f32 rasters); the numpy/HiGHS oracle it mirrors runs ~30 env-steps/s per core")


# ------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--tower_height", type=int, default=4)
    ap.add_argument("--max_steps", type=int, default=15)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--groups", type=int, default=2, help="independent env groups per GPU, one HIP stream each")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-launch HIP events (experiment: their cost)")
    ap.add_argument("--debug", type=int, default=0, help="kernel timing experiments (bit0: skip the LPs) -- invalidates the run")
    ap.add_argument("--no-f32-rasters", action="store_true", help="bit-packed rasters only (reported as its own mode)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the short runs of the two other raster modes")
    ap.add_argument("--sparse-raster-update", action="store_true",
                    help="f32 rasters, but only the row groups that change are stored (reported as its own mode)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.tower_height, args.max_steps)      # before the GPU is initialised (fork pool)

    import torch
    import torch.distributed as dist
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGymGroups

    # one process per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share one card to rehearse the N>1 code path
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    H = 0.8                                                     # gym_env.py:46 bridge_setup(H=.8, num_stories=N)
    targets = [(0.5, 0.0, args.tower_height * H + H / 2)]
    obstacles = [(0.5, 0.0, i * H + H / 2) for i in range(args.tower_height)]
    env = VecAssemblyGymGroups(args.envs, [load_urdf("shapes/trapezoid.urdf")], obstacles, targets,
                               groups=args.groups, max_steps=args.max_steps, seed=args.seed * 1000003 + rank,
                               device=dev, f32_rasters=not args.no_f32_rasters, debug=args.debug,
                               sparse_raster_update=args.sparse_raster_update)
    lockstep = env.lockstep_random

    for _ in range(args.warmup):
        lockstep()

    def sync():
        env.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    sync()
    s0 = env.read_stats()
    if not args.no_kernel_timing:
        env.timing_begin(args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lockstep()
    sync()
    dt = time.perf_counter() - t0
    raster_ms, n_launch = (0.0, 0) if args.no_kernel_timing else env.timing_end()
    s1 = env.read_stats()
    d = {k: s1[k] - s0[k] for k in s1}

    env_steps = float(d["env_steps"])
    if world > 1:
        t = torch.tensor([dt, env_steps], dtype=torch.float64, device=red_dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_all, steps_all = float(tmax[0]), float(tsum[1])
    else:
        dt_all, steps_all = dt, env_steps

    if rank == 0:
        units = args.envs * args.steps                      # env slots processed by the timed rasteriser launches
        n_launch = max(n_launch, 1)
        # HBM traffic per launch: measured/algorithmic ratio of the committed PMC passes (profiles/), applied to
        # this run's launch size -- PMC counters cannot be collected inside an un-profiled bench run
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_k_raster.json")))
            # counters were taken at 4096 envs in one group; scale to this run's launch size
            traffic_ratio = pm["hbm_bytes_per_launch"] / pm["algorithmic_bytes_per_launch"]
        except Exception:
            traffic_ratio = None
        alg = algorithmic_bytes(d["sum_cand"], d["sum_blocks"], units)
        per_launch = alg / max(n_launch, 1)
        avg_ms = raster_ms / max(n_launch, 1)
        achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        out = {
            "metric": "env-steps/sec (vectorised assembly_gym, tower_height=%d)" % args.tower_height,
            "value": steps_all / dt_all,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_all / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d envs/GPU lock-step, bridge_setup(num_stories=%d), trapezoid, max_steps=%d, "
                            "uniform-random policy, %s" % (args.envs, args.tower_height, args.max_steps,
                                                           "bit-packed rasters only" if args.no_f32_rasters
                                                           else "f32 64x64 rasters for every raw candidate, sparse row-group update"
                                                           if args.sparse_raster_update
                                                           else "f32 64x64 rasters for every raw candidate"),
                "envs_per_gpu": args.envs, "groups": args.groups, "tower_height": args.tower_height, "max_steps": args.max_steps,
                "mean_raw_candidates": d["sum_cand"] / max(units, 1),
                "mean_valid_candidates": d["sum_valid"] / max(units, 1),
                "mean_blocks": d["sum_blocks"] / max(units, 1),
                "env_step_fraction": env_steps / max(units, 1),
                "lp_errors": d["lp_errors"], "interface_overflows": d["if_overflow"],
                "bytes_per_env_step": alg / max(units, 1),
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_raster",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (traffic_ratio * per_launch) if traffic_ratio else None,
                "avg_launch_ms": avg_ms, "launches": n_launch, "algorithmic_bytes_per_launch": per_launch,
                "whole_step_GBps": alg / dt / 1e9,
            },
        }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and not (args.no_other_modes or args.no_f32_rasters or args.sparse_raster_update or args.debug):
            # the two other raster modes of the same workload, for orientation only: never part of `value`.  Each runs
            # in its own child process (as if started by hand): how HIP maps the streams of a SECOND set of groups onto
            # hardware queues inside one process moved these latency-bound modes by +-30 %
            try:
                del env, lockstep
                torch.cuda.empty_cache()
                import subprocess
                out["other_modes"] = {}
                for name, flag in (("sparse_raster_update", "--sparse-raster-update"),
                                   ("bit_packed_rasters_only", "--no-f32-rasters")):
                    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-other-modes", flag,
                           "--groups", "3", "--envs", str(args.envs), "--steps", str(args.steps), "--warmup", str(args.warmup),
                           "--tower_height", str(args.tower_height), "--max_steps", str(args.max_steps), "--seed", str(args.seed)]
                    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
                    line = [l for l in res.stdout.splitlines() if l.startswith("{")]
                    if res.returncode != 0 or not line:
                        out["other_modes"][name] = {"error": (res.stderr or "no output")[-200:]}
                        continue
                    sub = json.loads(line[-1])
                    out["other_modes"][name] = {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"],
                                                "groups": sub["config"]["groups"]}
            except Exception as exc:                     # the headline line must survive whatever happens here
                out["other_modes"] = {"error": repr(exc)[:200]}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
