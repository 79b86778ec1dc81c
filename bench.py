#!/usr/bin/env python3
"""bench.py -- env-steps/sec of the vectorised assembly_gym hot path (BASELINE.json metric).

One "step" of this bench = one lock-step of the hot path over one batch of E synthetic
environments: uniform-random policy draw -> place the block -> contact interfaces -> stability
{last frozen, none frozen} -> reward/termination/auto-reset -> enumerate + place the A raw
candidates -> (A+1) 64x64 f32 rasters -> bounds/overlap mask -> linear reward  (SURVEY.md §8d).
`value` counts real environment steps (reset-only lock-steps of an env are not counted).

  python bench.py [--gpus N --steps K --warmup W]

With --gpus N > 1 and no WORLD_SIZE in the environment the script starts its own N ranks (a
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` child,
before anything touches the GPU) and passes the child's output through; launched under torchrun
by someone else it is simply one of the ranks.

Envs are independent: each rank owns E envs (weak scaling), there is no data-path collective;
the only communication is the barrier / max-reduce of the timing contract.

Side numbers (never part of `value`) ride in the same JSON line under `other_modes`, each measured in
a child process of its own: the two cheaper raster modes, the candidate-stability mask
(`is_action_stable_rbe` over every valid candidate, LPs/s), BASELINE.json's config-5 simulator
workload (hexagon, bridge span) and training in the loop for configs[1], configs[2] and (one GPU's share of) configs[4].
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
PMC_FILE = os.path.join("profiles", "r04_pmc_k_raster.json")


def algorithmic_bytes(sum_cand, sum_blocks, n_envsteps_units, V=4, f32_rasters=True):
    """bytes one lock-step must move for `n_envsteps_units` environments holding
    sum_blocks blocks and sum_cand raw candidates in total (f32 rasters as the reference surfaces them; with
    f32_rasters=False the images exist only as 64 x u64 row masks, 512 B each)."""
    A, k, E = float(sum_cand), float(sum_blocks), float(n_envsteps_units)
    px = 4 * 64 * 64 if f32_rasters else 8 * 64
    return (px * (A + E)                   # action + state rasters written
            + 2 * px * E                   # state + obstacle rasters read for the overlap test
            + 8 * 2 * V * (k + A)          # vertex reads, f64
            + 24 * (k + A)                 # poses
            + 16 * A + A + 4 * A           # candidate descriptors, mask, lin_reward
            + 100 * E)                     # stability I/O


def task_of(args):
    """(shape names, obstacles, targets, label) of the synthetic task: --bridge_length N -> horizontal_bridge_setup
    (gym_env.py:25-43), else --tower_height N -> bridge_setup(H=.8, num_stories=N) (gym_env.py:46-61)."""
    names = dict(trapezoid=["trapezoid"], hexagon=["hexagon"], both=["trapezoid", "hexagon"])[args.shapes]
    if args.bridge_length:
        sq, n = 0.6, args.bridge_length
        targets = [(n * sq + 2.5 * sq, 0.0, sq / 2)]
        obstacles = [(i * sq, 0.0, sq / 2) for i in range(1, n + 1)]
        label = "horizontal_bridge_setup(num_obstacles=%d)" % n
    else:
        H, n = 0.8, args.tower_height
        targets = [(0.5, 0.0, n * H + H / 2)]
        obstacles = [(0.5, 0.0, i * H + H / 2) for i in range(n)]
        label = "bridge_setup(num_stories=%d)" % n
    return names, obstacles, targets, label


# ------------------------------------------------------------------------- CPU baseline (oracle)
def _oracle_setup(args):
    from oracle.env import bridge_setup, horizontal_bridge_setup
    trap, hexa = args.shapes in ("trapezoid", "both"), args.shapes in ("hexagon", "both")
    if args.bridge_length:
        return horizontal_bridge_setup(num_obstacles=args.bridge_length, trapezoid=trap, hexagon=hexa)
    return bridge_setup(num_stories=args.tower_height, trapezoid=trap, hexagon=hexa)


def _cpu_worker(job):
    seed, env_id, seconds, setup, max_steps = job
    from oracle.c_env import CEnv
    from oracle.env import OracleGym
    ce = CEnv(OracleGym(**setup, max_steps=max_steps))
    ce.enable_f32()                                  # same unit of work: f32 rasters for every raw candidate
    ce.run(seed, env_id, 200)                        # warm-up
    n, chunk = 0, 2000
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        n += ce.run(seed, env_id, chunk)
    return n, time.perf_counter() - t0


def _numpy_highs_worker(job):
    """The numpy + HiGHS restatement (oracle/env.py) itself, one process: the same lock-step protocol, rasters of
    every raw candidate, LPs by scipy.optimize.linprog."""
    seed, min_steps, seconds, setup, max_steps = job
    from oracle.env import OracleGym, OracleLockstep, policy_draw
    o = OracleLockstep(OracleGym(**setup, max_steps=max_steps))
    n, ctr = 0, [0]

    def pick(nv):
        r = policy_draw(seed, 0, ctr[0]) % nv
        ctr[0] += 1
        return r
    t0 = time.perf_counter()
    while n < min_steps or time.perf_counter() - t0 < seconds:
        n += 1 if o.lockstep(pick)["valid_step"] else 0
    return n, time.perf_counter() - t0


def cpu_baseline(args, seconds=10.0):
    """The plain-C restatement of the path (oracle/c, tested bit for bit against the numpy + HiGHS oracle), one
    environment per process on every host core (the headline `value`), then alone on one core, then the numpy +
    HiGHS oracle on one core (SURVEY.md §8d asks for all three)."""
    import multiprocessing as mp
    from oracle import c_env
    c_env.lib()                                      # build once before forking
    setup = _oracle_setup(args)
    # a one-GPU box of this pool gives the job a 16-core share whatever the affinity mask says
    seconds = float(os.environ.get("BENCH_CPU_SECONDS", seconds))          # tests shorten the sample
    cores = int(os.environ.get("BENCH_CPU_WORKERS", max(1, min(len(os.sched_getaffinity(0)), 16))))
    ctx = mp.get_context("fork")
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(0, i, seconds, setup, args.max_steps) for i in range(cores)])
        steps = sum(r[0] for r in res)
        wall = max(r[1] for r in res)
        one = pool.map(_cpu_worker, [(0, 0, max(1.0, seconds / 2), setup, args.max_steps)])[0]
        nh = pool.map(_numpy_highs_worker, [(0, int(os.environ.get("BENCH_NUMPY_STEPS", 200)), max(1.0, seconds / 2),
                                             setup, args.max_steps)])[0]
    return dict(value=steps / wall, unit="env-steps/s", cores=cores, kind="port",
                sample=f"{cores} processes x {seconds:.0f} s of {task_of(args)[3]} random-policy lock-steps "
                       f"({steps} env-steps) with oracle/c/oracle_env.c (scalar C, -O2, float64, own simplex, "
                       f"bit + f32 rasters)",
                one_core=dict(value=one[0] / one[1], unit="env-steps/s", cores=1, kind="port",
                              sample=f"the same C port, 1 process alone, {one[1]:.1f} s ({one[0]} env-steps)"),
                numpy_highs=dict(value=nh[0] / nh[1], unit="env-steps/s", cores=1, kind="port",
                                 sample=f"oracle/env.py (numpy float64 + scipy HiGHS LPs), 1 process, {nh[1]:.1f} s "
                                        f"({nh[0]} env-steps)"))


# ------------------------------------------------------------------------- self-launch for --gpus N
def self_launch(args):
    """--gpus N > 1 without a torchrun environment: start N ranks ourselves.  Runs before torch is imported, so this
    parent never touches the GPU; the ranks are children and the parent exits with their code."""
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.run(cmd, env=env).returncode


# ------------------------------------------------------------------------- side modes (child processes)
def _child_json(cmd, timeout, env=None):
    """Run a child job (its own process group, so that a time-out takes its grandchildren -- torchrun's ranks -- with it)
    and return (the last JSON line it printed, None) or (None, what went wrong)."""
    import signal
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)           # the exact process group started above
        except ProcessLookupError:
            pass
        proc.communicate()
        return None, f"timed out after {timeout} s"
    line = [l for l in out.splitlines() if l.startswith("{")]
    if proc.returncode != 0 or not line:
        return None, (err or "no output")[-300:]
    return json.loads(line[-1]), None


def side_modes(args):
    """Each in a process of its own (as if started by hand): how HIP maps the streams of a SECOND set of groups onto
    hardware queues inside one process moved the latency-bound modes by +-30 %."""
    me = os.path.abspath(__file__)
    # a side mode is one seed, so it gets at least 100 timed lock-steps behind 20 of warm-up whatever --steps / --warmup the
    # headline was given (20 lock-steps are 10-30 ms: one slow launch moved such a sample by 20 %); its own counts are reported
    side_steps, side_warmup = max(args.steps, 100), max(args.warmup, 20)
    base = [sys.executable, me, "--no-cpu-baseline", "--no-other-modes", "--envs", str(args.envs), "--steps", str(side_steps),
            "--warmup", str(side_warmup), "--seeds", str(args.seeds).split(",")[0]]
    task = ["--tower_height", str(args.tower_height), "--max_steps", str(args.max_steps), "--shapes", args.shapes]
    if args.bridge_length:
        task += ["--bridge_length", str(args.bridge_length)]
    out = {}

    def sim(name, extra):
        try:
            sub, err = _child_json(base + extra, 600)
            if sub is None:
                out[name] = {"error": err}
                return
            out[name] = {"value": sub["value"], "unit": sub["unit"], "ms_per_step": sub["ms_per_step"], "steps": sub["steps"],
                         "warmup": sub["warmup"], "groups": sub["config"].get("groups"), "workload": sub["config"]["workload"]}
            for k in ("roofline", "candidate_stability"):
                if k in sub:
                    out[name][k] = sub[k]
        except Exception as exc:
            out[name] = {"error": repr(exc)[:300]}

    sim("sparse_raster_update", task + ["--sparse-raster-update", "--groups", "3"])
    sim("bit_packed_rasters_only", task + ["--no-f32-rasters", "--groups", "3"])
    # the LP pass behind the chain of ONE group: its kernels alone on the chip (decisions_per_s = their own rate); then with
    # three groups, each group's LP pass beside the other groups' rasterisers (tools/cand_groups_sweep.sh: 2.76 / 3.01 / 3.18 M
    # env-steps/s for 1 / 2 / 3 groups -- the passes stretch beside a rasteriser that holds every wave slot, decisions_per_s_wall
    # is the figure to read there)
    sim("candidate_stability", task + ["--mode", "candidate-stability", "--groups", "1"])
    sim("candidate_stability_3groups", task + ["--mode", "candidate-stability", "--groups", "3"])
    # BASELINE.json configs[4]'s simulator workload: hexagon blocks, bridge-span task, max_steps=15
    # (three env groups: its rasteriser launches are twice as long as the tower task's, so the third group's overlap is
    # worth more than the extra ramp / drain -- tools/hex_groups_sweep.sh: 2.39 / 2.47 / 2.14 M env-steps/s for 2 / 3 / 4)
    sim("config5_hexagon_bridge", ["--shapes", "hexagon", "--bridge_length", "3", "--max_steps", "15", "--groups", "3"])
    if os.environ.get("BENCH_TRAIN_MODES", "1") != "0":
        tool = os.path.join(ROOT, "tools", "train_throughput.py")
        n_ls = os.environ.get("BENCH_TRAIN_LOCKSTEPS", "12")
        # every leg starts from 4096 (1024) freshly reset -- i.e. identical -- environments; the warm-up lock-steps let the
        # episodes (4-5 steps each) drift apart before anything is timed, so that the share of environments that are in the same
        # state (and share their candidate rows, VecDQN._rows) is the steady-state one; it is reported as rows_fed_fraction
        n_warm = os.environ.get("BENCH_TRAIN_WARMUP", "20")
        for name, extra in (("train_config3_successor_mlp", ["--envs", "4096", "--tower", "4", "--max_steps", "15", "--model",
                                                             "SuccessorMLP", "--loss", "mse_block_features", "--locksteps", n_ls,
                                                             "--warmup", n_warm]),
                            ("train_config2_convnet", ["--envs", "1024", "--tower", "2", "--max_steps", "10", "--model",
                                                       "ConvNet", "--loss", "mse_q_values", "--locksteps", n_ls, "--warmup", n_warm]),
                            # BASELINE.json configs[4] on one GPU
                            ("train_config5_unet_hexagon", ["--envs", "4096", "--max_steps", "15", "--model", "UNet", "--loss",
                                                            "mse_q_values+mse_block_features", "--shapes", "hexagon",
                                                            "--bridge_length", "3", "--locksteps", n_ls, "--warmup", n_warm])):
            try:
                sub, err = _child_json([sys.executable, tool, *extra], 900)
                if sub is None:
                    out[name] = {"error": err}
                    continue
                out[name] = {"value": sub["env_steps_per_s"], "unit": "env-steps/s (acting + replay + 25 optimiser steps per lock-step)",
                             "ms_per_lockstep": sub["ms_per_lockstep"], "ms_act": sub["ms_act"], "ms_targets": sub["ms_targets"],
                             "ms_per_train_step": sub["ms_per_train_step"],
                             # candidate rows of all envs per lock-step, and the share of them a network was actually fed: envs in
                             # the same state share one set of rows (exact; tools/train_throughput.py --no_dedup feeds all)
                             "rows_per_lockstep": sub.get("rows_per_lockstep"), "rows_fed_fraction": sub.get("rows_fed_fraction"),
                             "config": sub["config"]}
            except Exception as exc:
                out[name] = {"error": repr(exc)[:300]}
    return out


# ------------------------------------------------------------------------- N > 1: BASELINE.json configs[3]
def train_config4_leg(args, dev, rank, world, backend):
    """configs[3] ("8 x MI355X over xGMI, 4096 envs/GPU, tower_height=4, mse_q_values+mse_block_features, shared replay via
    RCCL all-gather") on the ranks of this run: every rank steps its own env shard, ONE all_gather_into_tensor of the
    888-B transition records per lock-step fills every rank's replica of the replay ring, every rank takes the same 25
    optimiser steps (robotoddler/training/vec_dqn.py, distributed.py).  Reported beside the headline, never part of
    `value`.  Collective: the process group bench.py was started with (nccl = RCCL on a node; gloo in the rehearsal)."""
    import hashlib
    import torch
    import torch.distributed as dist
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGym
    from robotoddler.training import distributed as D
    from robotoddler.training.successor_dqn import build_parser, make_nets
    from robotoddler.training.vec_dqn import VecDQN
    loss = "mse_q_values+mse_block_features"
    E = int(os.environ.get("BENCH_CONFIG4_ENVS", args.envs))
    n_ls, n_warm, n_train = int(os.environ.get("BENCH_TRAIN_LOCKSTEPS", "12")), int(os.environ.get("BENCH_TRAIN_WARMUP", "20")), 25
    targs = vars(build_parser().parse_args(["--model", "SuccessorMLP", "--loss_function", loss, "--learning_rate", "1e-4"]))
    torch.manual_seed(0)                                          # identical initial weights on every rank
    pol, tgt = make_nets(targs, dev)
    H, n = 0.8, 4
    env = VecAssemblyGym(E, [load_urdf("shapes/trapezoid.urdf")], [(0.5, 0.0, i * H + H / 2) for i in range(n)],
                         [(0.5, 0.0, n * H + H / 2)], max_steps=15, seed=rank, device=dev, env_id_base=rank * E,
                         f32_rasters=VecDQN.acting_needs_f32_rasters(pol), candidate_snapshots=False)
    opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
    agent = VecDQN(pol, tgt, opt, env, max(2000, 4 * E * world), 32, 0.95, 0.01, loss, seed=0, rank=rank)
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    def sync():
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(n_warm):
        agent.lockstep(n_train)
    sync()
    steps0, t0 = agent.env_steps, time.perf_counter()
    pending, rec = None, None
    for _ in range(n_ls):
        losses, rec_all = agent.lockstep(n_train, defer_losses=True)
        if pending is not None:
            pending.get()
        pending = losses
    last = pending.get() if pending is not None else []
    sync()
    dt = time.perf_counter() - t0
    # the collective alone, on a full-size payload (every env valid), bracketed by device syncs
    rec = torch.zeros((E, agent.ring.data.shape[1]), dtype=torch.float64, device=dev)
    valid = torch.ones(E, dtype=torch.bool, device=dev)
    for _ in range(3):
        D.all_gather_records(rec, valid)
    sync()
    t1 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        got = D.all_gather_records(rec, valid)
    torch.cuda.synchronize()
    ag_ms = (time.perf_counter() - t1) / reps * 1e3
    ring = agent.ring
    order = (ring.head - ring.size + torch.arange(ring.size, device=ring.data.device)) % ring.capacity
    h_ring = hashlib.sha256(ring.data[order].cpu().numpy().tobytes()).digest()[:8]
    h_pol = hashlib.sha256(pol._flat_params.flat.detach().cpu().numpy().tobytes()).digest()[:8]
    mine = torch.tensor([int.from_bytes(h_ring, "little", signed=True), int.from_bytes(h_pol, "little", signed=True),
                         agent.env_steps - steps0, int(got.shape[0])], dtype=torch.int64, device=red_dev)
    allv = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allv, mine)
    tt = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    allv = torch.stack(allv).cpu()
    steps_all = int(allv[:, 2].sum())
    return {"value": steps_all / float(tt[0]), "unit": "env-steps/s (all ranks: acting + record all-gather + replay + 25 optimiser steps per lock-step)",
            "config": "BASELINE.json configs[3] on %d rank(s): %d envs/rank, tower_height=4, SuccessorMLP, %s, replicated "
                      "replay ring filled by one all_gather_into_tensor of %d-B records per lock-step"
                      % (world, E, loss, 8 * agent.ring.data.shape[1]),
            "ms_per_lockstep": float(tt[0]) / n_ls * 1e3, "locksteps": n_ls, "ranks_seen": dist.get_world_size(),
            "dist_backend": backend, "allgather_ms_per_lockstep": ag_ms,
            "allgather_bytes_per_rank": int(E * (agent.ring.data.shape[1] + 1) * 8),
            "allgather_rows_received": int(allv[0, 3]),
            "ring_records": int(ring.size), "ring_hash_equal": bool((allv[:, 0] == allv[0, 0]).all()),
            "policy_hash_equal": bool((allv[:, 1] == allv[0, 1]).all()),
            "last_losses_finite": bool(all(l == l and l >= 0.0 for l in last))}


# ------------------------------------------------------------------------- main
def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--tower_height", type=int, default=4)
    ap.add_argument("--bridge_length", type=int, default=0, help="> 0: horizontal_bridge_setup(num_obstacles=N) instead of the tower")
    ap.add_argument("--shapes", choices=["trapezoid", "hexagon", "both"], default="trapezoid")
    ap.add_argument("--max_steps", type=int, default=15)
    ap.add_argument("--seeds", type=str, default="0,1,2",
                    help="policy seeds, one full measurement (W warm-up + K timed lock-steps) each; `value` = the median run")
    ap.add_argument("--groups", type=int, default=2, help="independent env groups per GPU, one HIP stream each")
    ap.add_argument("--mode", choices=["sim", "candidate-stability"], default="sim",
                    help="candidate-stability: every lock-step also decides is_action_stable_rbe for every valid candidate")
    ap.add_argument("--snapshots", action="store_true",
                    help="keep the per-env tableau snapshots of the candidate-stability kernel in the simulator modes too (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the per-launch HIP events (experiment: their cost)")
    ap.add_argument("--debug", type=int, default=0,
                    help="kernel timing experiments of a diagnostic build (BRIDGES_LIB=tools/libbridges_hip_diag.so) -- invalidates the run")
    ap.add_argument("--no-f32-rasters", action="store_true", help="bit-packed rasters only (reported as its own mode)")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the child runs of the side modes")
    ap.add_argument("--config4-leg", action="store_true",
                    help="(internal) run only BASELINE configs[3]'s training leg on the ranks of this job and print its JSON")
    ap.add_argument("--sparse-raster-update", action="store_true",
                    help="f32 rasters, but only the row groups that change are stored (reported as its own mode)")
    return ap.parse_args()


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start one rank per GPU")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and args.mode == "sim":
        cpu = cpu_baseline(args)                       # before the GPU is initialised (fork pool)

    import torch
    import torch.distributed as dist
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGym, VecAssemblyGymGroups

    # one process per GPU; BENCH_DIST_BACKEND=gloo lets several ranks share one card to rehearse the N>1 code path
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    ranks_seen = 1
    # BRIDGES_FORCE_COLLECTIVE=1: a one-rank job keeps its process group, so that every collective of the N > 1 path (RCCL
    # all-gather of the records, the timing all-reduces, barriers) executes on a one-GPU box too
    use_pg = world > 1 or os.environ.get("BRIDGES_FORCE_COLLECTIVE", "0") == "1"
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:          # started without torchrun
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)          # RCCL over xGMI
        else:
            dist.init_process_group(backend)
        ranks_seen = dist.get_world_size()
    red_dev = dev if backend == "nccl" else torch.device("cpu")

    if args.config4_leg:                                # child job of an N > 1 run: only the training leg, its own process group
        leg = train_config4_leg(args, dev, rank, world, backend)
        if rank == 0:
            print(json.dumps(leg), flush=True)
        if use_pg:
            dist.destroy_process_group()
        return

    names, obstacles, targets, task_label = task_of(args)
    geoms = [load_urdf(f"shapes/{n}.urdf") for n in names]
    V = sum(g.num_faces_2d for g in geoms) / len(geoms)
    cand_mode = args.mode == "candidate-stability"
    seeds = [int(v) for v in str(args.seeds).split(",") if v != ""]

    def measure(seed):
        """W untimed + exactly K timed lock-steps of a freshly created environment set with policy seed `seed`.  The policy
        RNG stream of an env is keyed by (seed, GLOBAL env id = rank * envs + e): an N-rank run steps the same
        trajectories as one run over N * envs environments."""
        # the per-env "last block frozen" tableau snapshots only serve candidate_stability_mask(): kept in that mode only
        kw = dict(max_steps=args.max_steps, seed=seed, device=dev, f32_rasters=not args.no_f32_rasters,
                  debug=args.debug, sparse_raster_update=args.sparse_raster_update,
                  candidate_snapshots=cand_mode or args.snapshots, env_id_base=rank * args.envs)
        cand_ev, cand_count = [], []
        if cand_mode:
            # every group: lock-step, then the LP pass over its valid candidates, on the group's stream -- the latency-bound LPs
            # of one group run beside the rasteriser of the next (one group, LPs behind the chain: 1.49 ms per lock-step)
            env = VecAssemblyGymGroups(args.envs, geoms, obstacles, targets, groups=args.groups, **kw)

            def lockstep():
                if len(cand_ev) < 4096 * env.G:
                    timed = []
                    env.lockstep_random_candidates(timed)
                    for a, b, n_dec in timed:
                        cand_ev.append((a, b))
                        cand_count.append(n_dec)
                else:
                    env.lockstep_random_candidates()
        else:
            env = VecAssemblyGymGroups(args.envs, geoms, obstacles, targets, groups=args.groups, **kw)
            lockstep = env.lockstep_random

        for _ in range(args.warmup):
            lockstep()

        def sync():
            env.sync()
            torch.cuda.synchronize()
            if use_pg:
                dist.barrier()
            torch.cuda.synchronize()

        sync()
        cand_ev.clear()
        cand_count.clear()
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
        if use_pg:
            t = torch.tensor([dt, env_steps], dtype=torch.float64, device=red_dev)
            tmax = t.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            tsum = t.clone()
            dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
            dt_all, steps_all = float(tmax[0]), float(tsum[1])
        else:
            dt_all, steps_all = dt, env_steps
        return dict(seed=seed, env=env, dt=dt, d=d, env_steps=env_steps, dt_all=dt_all, steps_all=steps_all,
                    raster_ms=raster_ms, n_launch=n_launch, cand_ev=cand_ev, cand_count=cand_count)

    # seeds {0, 1, 2}; `value` is the MEDIAN run, every run's value is listed in config.per_seed
    runs = []
    for sd in seeds:
        r = measure(sd)
        runs.append(r)
        if sd != seeds[-1]:
            r["env"] = None                            # free the 20 GB of buffers before the next set is created
            torch.cuda.empty_cache()
    order = sorted(range(len(runs)), key=lambda i: runs[i]["steps_all"] / runs[i]["dt_all"])
    med = runs[order[len(order) // 2]]
    env = runs[-1]["env"]                              # (side figures of the candidate-stability mode read the last set)
    dt, d, env_steps, dt_all, steps_all = med["dt"], med["d"], med["env_steps"], med["dt_all"], med["steps_all"]
    raster_ms, n_launch, cand_ev, cand_count = med["raster_ms"], med["n_launch"], runs[-1]["cand_ev"], runs[-1]["cand_count"]

    def release():
        """Drop every environment set (18.8 GB of raster buffers each) before a side mode allocates its own."""
        nonlocal env
        for r in runs:
            r["env"] = None
        env = None
        torch.cuda.empty_cache()

    if rank == 0:
        units = args.envs * args.steps                      # env slots processed by the timed rasteriser launches
        n_launch = max(n_launch, 1)
        # HBM traffic per launch: measured/algorithmic ratio of the committed PMC passes (profiles/), applied to
        # this run's launch size -- PMC counters cannot be collected inside an un-profiled bench run
        try:
            pm = json.load(open(os.path.join(ROOT, PMC_FILE)))
            traffic_ratio = pm["hbm_bytes_per_launch"] / pm["algorithmic_bytes_per_launch"]
        except Exception:
            traffic_ratio = None
        alg = algorithmic_bytes(d["sum_cand"], d["sum_blocks"], units, V, f32_rasters=not args.no_f32_rasters)
        per_launch = alg / max(n_launch, 1)
        avg_ms = raster_ms / max(n_launch, 1)
        achieved = per_launch / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        raster_mode = ("bit-packed rasters only" if args.no_f32_rasters
                       else "f32 64x64 rasters for every raw candidate, sparse row-group update" if args.sparse_raster_update
                       else "f32 64x64 rasters for every raw candidate")
        if cand_mode:
            raster_mode += ", is_action_stable_rbe for every valid candidate"
        if args.bridge_length:
            metric = "env-steps/sec (vectorised assembly_gym, %s, %s)" % (task_label, args.shapes)
        else:
            metric = "env-steps/sec (vectorised assembly_gym, tower_height=%d)" % args.tower_height
        out = {
            "metric": metric,
            "value": steps_all / dt_all,
            "unit": "env-steps/s",
            "n_gpus": world,
            "ranks_seen": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt_all / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d envs/GPU lock-step, %s, %s, max_steps=%d, uniform-random policy, %s"
                            % (args.envs, task_label, args.shapes, args.max_steps, raster_mode),
                "envs_per_gpu": args.envs, "groups": env.G,
                "seeds": seeds, "seed_of_value": med["seed"],
                "per_seed": [{"seed": r["seed"], "value": r["steps_all"] / r["dt_all"], "ms_per_step": r["dt_all"] / args.steps * 1e3}
                             for r in runs],
                "env_ids": "policy RNG keyed by (seed, global env id = rank * envs_per_gpu + e)",
                "tower_height": args.tower_height,
                "bridge_length": args.bridge_length, "shapes": args.shapes, "max_steps": args.max_steps,
                "mean_raw_candidates": d["sum_cand"] / max(units, 1),
                "mean_valid_candidates": d["sum_valid"] / max(units, 1),
                "mean_blocks": d["sum_blocks"] / max(units, 1),
                "env_step_fraction": env_steps / max(units, 1),
                "lp_errors": d["lp_errors"], "interface_overflows": d["if_overflow"], "cand_overflow": d["cand_overflow"],
                "bytes_per_env_step": alg / max(units, 1),
                "debug": args.debug, "tableau_snapshots": bool(cand_mode or args.snapshots),
                "dist_backend": backend if use_pg else None,
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_raster",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": (traffic_ratio * per_launch) if traffic_ratio else None,
                "traffic_source": "static ratio (HBM bytes / algorithmic bytes of the committed PMC passes, %s) applied to this "
                                  "run's algorithmic bytes per launch; not measured in this run" % PMC_FILE,
                "avg_launch_ms": avg_ms, "launches": n_launch, "algorithmic_bytes_per_launch": per_launch,
                "whole_step_GBps": alg / dt / 1e9,
            },
        }
        if args.no_f32_rasters:
            out["roofline"]["traffic"] = None        # the PMC ratio was taken on the f32 rasteriser
            out["roofline"]["note"] = ("bit-packed mode: 512 B per image, the rasteriser is bound by its f64 half-plane "
                                       "tests, not by HBM; frac is reported for completeness only")
        if args.sparse_raster_update:
            # the kernel stores only the row groups that hold or held pixels, so bytes-of-the-full-rewrite over its
            # launch time is not a bandwidth: no roofline figure is claimed for this mode
            out["roofline"] = {"bound": "hbm", "kernel": "k_raster", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": None, "traffic": None, "avg_launch_ms": avg_ms, "launches": n_launch,
                               "note": "sparse row-group update stores fewer bytes than the full rewrite the §8(d) "
                                       "formula counts; stored bytes are not counted in this run"}
        if cand_mode:
            torch.cuda.synchronize()
            ms = sum(a.elapsed_time(b) for a, b in cand_ev)
            n_dec = int(sum(int(c) for c in cand_count))
            g0 = env.envs[-1]                           # verdict counts of the last lock-step: one group's candidates
            total = g0.total_candidates()
            cst = g0.cand_stable[:total][g0.cand_mask[:total].bool()]
            n_ls = max(len(cand_ev) // env.G, 1)
            out["candidate_stability"] = {
                # decisions_per_s: over the LP passes' own durations (each group's pass timed on its stream, beside whatever the
                # other groups run); decisions_per_s_wall: over the wall time of the whole run (simulator included)
                "decisions": n_dec, "decisions_per_s": n_dec / (ms * 1e-3) if ms > 0 else 0.0, "unit": "LPs/s",
                "decisions_per_s_wall": n_dec / runs[-1]["dt"] if runs[-1]["dt"] > 0 else 0.0,
                "ms_per_lockstep": ms / n_ls, "decisions_per_lockstep": n_dec / n_ls, "groups": env.G,
                "last_lockstep": {"stable": int((cst == 1).sum()), "unstable": int((cst == 0).sum()), "errors": int((cst == 2).sum()),
                                  "queued_large_tableaux": int(g0.cand_counters[0])}}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if use_pg:
            release()
        plain = not (args.no_f32_rasters or args.sparse_raster_update or args.debug or cand_mode or args.bridge_length
                     or args.shapes != "trapezoid")
        if world == 1 and plain and not args.no_other_modes:
            try:
                release()
                out["other_modes"] = side_modes(args)
            except Exception as exc:                     # the headline line must survive whatever happens here
                out["other_modes"] = {"error": repr(exc)[:200]}
    # BASELINE.json configs[3] on the same N ranks, as a CHILD job with its own process group and a timeout: the
    # collectives of the training loop have never run on more than one physical GPU from inside this build, and a hang or
    # a crash in there must not cost the headline line.  Every rank of this job leaves its GPU first.
    want_leg = use_pg and args.mode == "sim" and os.environ.get("BENCH_TRAIN_MODES", "1") != "0"
    if use_pg:
        release()
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if want_leg:
            try:
                with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                    sk.bind(("127.0.0.1", 0))
                    port = sk.getsockname()[1]
                cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                       "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), "--config4-leg",
                       "--gpus", str(world), "--envs", str(args.envs)]
                env_c = {k: v for k, v in os.environ.items()
                         if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE",
                                      "ROLE_WORLD_SIZE", "GROUP_WORLD_SIZE", "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT",
                                      "TORCHELASTIC_MAX_RESTARTS")}
                env_c.update(MASTER_ADDR="127.0.0.1")
                leg, err = _child_json(cmd, int(os.environ.get("BENCH_CONFIG4_TIMEOUT", "600")), env=env_c)
                if leg is None:
                    leg = {"error": err}
            except Exception as exc:                     # the headline line must survive whatever happens here
                leg = {"error": repr(exc)[:300]}
            out.setdefault("other_modes", {})["train_config4"] = leg
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
