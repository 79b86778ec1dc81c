#!/bin/bash
out=gpurun_out/r3h; mkdir -p $out
B="bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-other-modes --seeds 0"
run() { BRIDGES_RASTER_GATE=$1 BRIDGES_RASTER_GATE_DEPTH=$2 timeout -k 10 200 python $B --groups $3 2>$out/err | grep "^{" > $out/b.json && python - <<PY
import json; d=json.load(open("$out/b.json")); print("gate $1 depth $2 g$3 ms/step %.3f raster %.3f ms TB/s %.2f value %.0f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"]/1e3, d["value"]))
PY
}
for rep in 1 2; do
run 1 1 2; run 1 2 3; run 1 2 4; run 1 3 4; run 0 1 3; run 0 1 4; run 1 2 6
done
