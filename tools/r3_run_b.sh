#!/bin/bash
set -o pipefail
out=gpurun_out/r3b
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_full_size.py tests/test_gpu_api_golden.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -5 $out/tests.log
[ $rc -eq 0 ] || exit $rc
B="bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-modes"
for g in 2 1 3; do
  timeout -k 10 200 python $B --groups $g 2>$out/bench_g$g.err | grep "^{" > $out/bench_g$g.json && python - <<PY
import json; d=json.load(open("$out/bench_g$g.json")); print("g$g ms/step %.3f raster %.3f ms TB/s %.2f value %.0f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"]/1e3, d["value"]))
PY
done
timeout -k 10 200 python $B --groups 3 --no-f32-rasters 2>$out/bench_bits.err | grep "^{" > $out/bench_bits.json && python - <<PY
import json; d=json.load(open("$out/bench_bits.json")); print("bits-only g3 ms/step %.3f raster %.3f ms value %.0f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["value"]))
PY
timeout -k 10 200 python $B --groups 2 --shapes hexagon --bridge_length 3 2>$out/bench_hex.err | grep "^{" > $out/bench_hex.json && python - <<PY
import json; d=json.load(open("$out/bench_hex.json")); print("hex g2 ms/step %.3f raster %.3f ms TB/s %.2f value %.0f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"]/1e3, d["value"]))
PY
echo done
