#!/bin/bash
# round-3 probe A: where does the rasteriser lose bandwidth beside the other group's chain?
set -o pipefail
out=gpurun_out/r3a
mkdir -p $out
run() { name=$1; shift; timeout -k 10 200 python "$@" 2>$out/$name.err | grep "^{" > $out/$name.json || { echo "$name failed"; tail -3 $out/$name.err; return 1; }; }
B="bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-modes"
for dbg in 0 1 2 4 6; do
  run bench_g2_d$dbg $B --groups 2 --debug $dbg && python - <<PY
import json; d=json.load(open("$out/bench_g2_d$dbg.json")); print("g2 debug $dbg ms/step %.3f raster %.3f ms TB/s %.2f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"]/1e3))
PY
done
for g in 1 3; do
  run bench_g${g}_d0 $B --groups $g && python - <<PY
import json; d=json.load(open("$out/bench_g${g}_d0.json")); print("g$g ms/step %.3f raster %.3f ms TB/s %.2f"%(d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["achieved"]/1e3))
PY
done
P="tools/deferred_expand_probe.py"
for g in 1 2 3; do
  run probe_g${g} $P --groups $g && cat $out/probe_g${g}.json
  run probe_g${g}_noexp $P --groups $g --no-expand && cat $out/probe_g${g}_noexp.json
done
run probe_exponly $P --groups 1 --expand-only && cat $out/probe_exponly.json
run probe_exponly2 $P --groups 2 --expand-only && cat $out/probe_exponly2.json
echo done
