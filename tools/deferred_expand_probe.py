#!/usr/bin/env python3
"""Probe (round 3): what would ONE all-env f32 expansion launch per lock-step cost beside the bit-level chain?

The bit-packed simulator (f32_rasters=False: k_step -> k_scan -> k_enumerate -> k_raster(bits, mask, lin) -> k_select)
runs on its group streams; behind every group's lock-step an expansion launch (bridges_bits_to_f32, one wave per image,
pure 16 KiB stores from the 512-B row masks) goes to ONE extra stream, ordered behind the group's chain by an event.
The chain of the next lock-step does not wait for it (timing probe only: cand_bits is not double-buffered here).

  python tools/deferred_expand_probe.py [--envs 4096] [--groups 1] [--steps 100]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")):
    sys.path.insert(0, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--groups", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-expand", action="store_true")
    ap.add_argument("--expand-only", action="store_true")
    args = ap.parse_args()
    import torch
    from bridges_hip import abi
    from bridges_hip.shapes import load_urdf
    from bridges_hip.vec_env import VecAssemblyGymGroups
    dev = torch.device("cuda:0")
    H, n = 0.8, 4
    targets = [(0.5, 0.0, n * H + H / 2)]
    obstacles = [(0.5, 0.0, i * H + H / 2) for i in range(n)]
    env = VecAssemblyGymGroups(args.envs, [load_urdf("shapes/trapezoid.urdf")], obstacles, targets, groups=args.groups,
                               max_steps=15, seed=0, device=dev, f32_rasters=False, candidate_snapshots=False)
    L = abi.lib()
    per = [int(e.E * 62.3) for e in env.envs]                    # mean raw candidates + the state raster, per group
    out = [torch.empty((n_, 64, 64), dtype=torch.float32, device=dev) for n_ in per]
    xs = torch.cuda.Stream(device=dev)
    evs = [torch.cuda.Event() for _ in env.envs]
    tev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps * args.groups)]
    k = [0]

    def lockstep(timed):
        for g, (e, st) in enumerate(zip(env.envs, env.streams)):
            if not args.expand_only:
                rc = L.bridges_env_lockstep_random(e._env, C.c_void_p(st.cuda_stream))
                assert rc == 0
                evs[g].record(st)
            if not args.no_expand:
                if not args.expand_only:
                    xs.wait_event(evs[g])
                if timed:
                    tev[k[0]][0].record(xs)
                rc = L.bridges_bits_to_f32(per[g], C.c_void_p(e.cand_bits.data_ptr()), C.c_void_p(out[g].data_ptr()),
                                           C.c_void_p(xs.cuda_stream))
                assert rc == 0
                if timed:
                    tev[k[0]][1].record(xs)
                    k[0] += 1

    for _ in range(args.warmup):
        lockstep(False)
    env.sync(); xs.synchronize(); torch.cuda.synchronize()
    s0 = env.read_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lockstep(True)
    env.sync(); xs.synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = env.read_stats()
    ems = [a.elapsed_time(b) for a, b in tev[:k[0]]]
    steps = s1["env_steps"] - s0["env_steps"]
    gb = sum(per) * 16384 / 1e9
    print(json.dumps({"groups": args.groups, "expand": not args.no_expand, "expand_only": args.expand_only,
                      "ms_per_lockstep": dt / args.steps * 1e3, "env_steps_per_s": steps / dt,
                      "expand_ms_avg": sum(ems) / max(len(ems), 1), "expand_GB_per_lockstep": gb,
                      "expand_TBps_in_launch": (gb / args.groups) / (sum(ems) / max(len(ems), 1)) / 1e0 if ems else None,
                      "whole_TBps": gb / (dt / args.steps * 1e3)}))


if __name__ == "__main__":
    main()
