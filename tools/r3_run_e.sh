#!/bin/bash
set -o pipefail
out=gpurun_out/r3e; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_env_parity.py tests/test_gpu_full_size.py tests/test_gpu_api_golden.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -eq 0 ] || { tail -40 $out/tests.log; exit $rc; }
timeout -k 10 600 python tests/stress/stress_parity.py --envs 512 --locksteps 100 --task hexbridge --seed 7 | tail -1
B="bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-modes --seeds 0"
for i in 1 2; do
timeout -k 10 200 python $B --groups 2 --shapes hexagon --bridge_length 3 2>$out/bench_hex.err | grep "^{" > $out/bench_hex.json && python - <<PY
import json; d=json.load(open("$out/bench_hex.json")); print("hex g2 ms/step %.3f raster %.3f ms TB/s %.2f value %.0f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"]/1e3, d["value"]))
PY
done
