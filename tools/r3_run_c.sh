#!/bin/bash
set -o pipefail
out=gpurun_out/${1:-r3c}
mkdir -p $out
timeout -k 10 1500 python -m pytest tests -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -8 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -1; timeout -k 10 300 python tools/single_env_throughput.py --count_syncs > $out/single_env.json 2> $out/single_env.err; cut -c1-300 $out/single_env.json
timeout -k 10 900 python bench.py > $out/bench_default.json 2> $out/bench_default.err; rc=$?
python - <<PY
import json
d=json.load(open("$out/bench_default.json"))
print("value %.0f ms/step %.3f frac %.3f" % (d["value"], d["ms_per_step"], d["roofline"]["frac"]), d["config"]["per_seed"])
for k,v in d.get("other_modes",{}).items(): print(k, {a:b for a,b in v.items() if a in ("value","ms_per_step","ms_per_lockstep","ms_act","ms_targets","ms_per_train_step","error")})
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"])
PY
exit $rc
