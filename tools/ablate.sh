#!/bin/bash
# raster ablation: debug bit1 = skip the row runs of the rasteriser, bit2 = skip the f32 stores.  The switches exist only in the
# diagnostic build: run tools/build_diag.sh first (this script points BRIDGES_LIB at its output).
export BRIDGES_LIB=${BRIDGES_LIB:-$(dirname "$0")/libbridges_hip_diag.so}
for dbg in 0 2 4 6; do
  timeout -k 10 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --groups 1 --debug $dbg 2>&1 | grep "^{" > /tmp/ab.json
  python -c "import json; d=json.load(open('/tmp/ab.json')); print('debug', $dbg, 'ms/step', round(d['ms_per_step'],3), 'raster ms', round(d['roofline']['avg_launch_ms'],3), 'meanA', round(d['config']['mean_raw_candidates'],1))"
done
