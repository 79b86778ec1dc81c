#!/bin/bash
# configs[4]'s simulator workload (hexagon, bridge span 3, max_steps 15) against the number of env groups per GPU
set -o pipefail
mkdir -p gpurun_out/hexsweep
for g in 2 3 4; do
  timeout -k 10 240 python bench.py --shapes hexagon --bridge_length 3 --max_steps 15 --groups $g --seeds 0 --no-other-modes --no-cpu-baseline > gpurun_out/hexsweep/g$g.json 2> gpurun_out/hexsweep/g$g.err || { tail -5 gpurun_out/hexsweep/g$g.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/hexsweep/g$g.json").read().strip().splitlines()[-1])
print("groups $g", round(d["value"]), round(d["ms_per_step"],3), d.get("roofline",{}).get("frac"), flush=True)
PY
done
