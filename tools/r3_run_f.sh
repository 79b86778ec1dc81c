#!/bin/bash
set -o pipefail
out=gpurun_out/r3f
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_full_size.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -6 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tests/stress/stress_candidate_stability.py > $out/stress_cand.txt 2>&1; tail -5 $out/stress_cand.txt
timeout -k 10 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-modes --mode candidate-stability --seeds 0 2>$out/cand.err | grep "^{" > $out/cand.json
python - <<PY
import json; d=json.load(open("$out/cand.json")); print("cand: value %.0f ms/step %.3f"%(d["value"], d["ms_per_step"]), d["candidate_stability"])
PY
