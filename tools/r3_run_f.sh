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
B="bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-modes --seeds 0"
timeout -k 10 200 python $B --groups 2 --shapes hexagon --bridge_length 3 2>$out/bench_hex.err | grep "^{" > $out/bench_hex.json && python - <<PY
import json; d=json.load(open("$out/bench_hex.json")); print("hex g2 ms/step %.3f raster %.3f ms value %.0f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
PY
timeout -k 10 200 python $B --groups 2 2>$out/bench_g2.err | grep "^{" > $out/bench_g2.json && python - <<PY
import json; d=json.load(open("$out/bench_g2.json")); print("g2 ms/step %.3f raster %.3f ms TB/s %.2f value %.0f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"]/1e3, d["value"]))
PY
timeout -k 10 200 python $B --groups 3 --no-f32-rasters 2>$out/bench_bits.err | grep "^{" > $out/bench_bits.json && python - <<PY
import json; d=json.load(open("$out/bench_bits.json")); print("bits-only g3 ms/step %.3f value %.0f"%(d["ms_per_step"], d["value"]))
PY
