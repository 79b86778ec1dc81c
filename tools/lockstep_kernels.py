#!/usr/bin/env python3
"""Per-lock-step kernel list from a rocprofv3 kernel trace (`--kernel-trace --output-format csv`).

The trace is cut at the start of every `k_step` launch of the ROLLOUT environment (the launch with the largest grid:
the replay scratch env never steps); the last `--last` complete segments are averaged.  Prints kernel name, launches
per lock-step, microseconds per lock-step, sorted by time, plus the GPU-busy share of the segment.

  python tools/lockstep_kernels.py gpurun_out/prof_x_mlp [--last 4] [--top 40]
"""
import argparse, csv, glob, os, re, sys
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--last", type=int, default=4)
ap.add_argument("--top", type=int, default=45)
ap.add_argument("--cut", default="k_step")
ap.add_argument("--width", type=int, default=100, help="characters of the kernel name kept (torch functor names are long)")
a = ap.parse_args()
f = sorted(glob.glob(os.path.join(a.dir, "**", "*kernel_trace.csv"), recursive=True))
if not f:
    sys.exit("no kernel_trace.csv under " + a.dir)
rows = list(csv.DictReader(open(f[0])))
ks = lambda r: (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [int(r["Start_Timestamp"]) for r in rows if a.cut in r["Kernel_Name"]]
if len(cuts) < a.last + 1:
    sys.exit(f"only {len(cuts)} '{a.cut}' launches in the trace")
lo, hi = cuts[-(a.last + 1)], cuts[-1]
agg, cnt, busy = defaultdict(float), defaultdict(int), 0
last_end = lo
for r in rows:
    s, e = ks(r)
    if s < lo or s >= hi:
        continue
    name = r["Kernel_Name"] if a.width > 100 else re.sub(r"\(.*", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)[:a.width]
    agg[name] += (e - s) / 1e3
    cnt[name] += 1
    busy += max(0, e - max(s, last_end))
    last_end = max(last_end, e)
n = a.last
seg_us = (hi - lo) / 1e3 / n
print(f"# {f[0]}: last {n} lock-steps, {seg_us:.0f} us per lock-step wall, GPU busy {busy / 1e3 / n:.0f} us "
      f"({100 * busy / (hi - lo):.0f} %), {sum(cnt.values()) / n:.0f} launches per lock-step")
W = a.width
print(f"{'kernel':{W}s} {'launches':>8s} {'us':>9s} {'%':>6s}")
tot = sum(agg.values())
for name, us in sorted(agg.items(), key=lambda kv: -kv[1])[:a.top]:
    print(f"{name:{W}s} {cnt[name] / n:8.1f} {us / n:9.1f} {100 * us / tot:6.1f}")
small = sum(c for k, c in cnt.items() if agg[k] / c < 8.0) / n
print(f"# launches shorter than 8 us on average: {small:.0f} per lock-step")
