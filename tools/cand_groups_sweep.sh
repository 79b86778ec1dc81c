#!/bin/bash
# the candidate-stability mode (lock-step + is_action_stable_rbe for every valid candidate) against the number of env groups per GPU
set -o pipefail
mkdir -p gpurun_out/candsweep
for g in 1 2 3; do
  timeout -k 10 240 python bench.py --mode candidate-stability --groups $g --seeds 0 --no-other-modes --no-cpu-baseline > gpurun_out/candsweep/g$g.json 2> gpurun_out/candsweep/g$g.err || { tail -5 gpurun_out/candsweep/g$g.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/candsweep/g$g.json").read().strip().splitlines()[-1])
c=d["candidate_stability"]
print("groups $g", round(d["value"]), "env-steps/s", round(d["ms_per_step"],3), "ms | LPs/s", round(c["decisions_per_s"]/1e6,1), "M (pass time)", round(c["decisions_per_s_wall"]/1e6,1), "M (wall) | pass ms", round(c["ms_per_lockstep"],3), c["last_lockstep"], flush=True)
PY
done
