#!/bin/bash
# A/B of the rasteriser stream arrangements of VecAssemblyGymGroups (BRIDGES_RASTER_STREAM):
# "" = raster gate between the group streams (default), "plain" = one extra stream for all rasterisers,
# "mask:<n>:<all|rest>" = group streams limited to the first n CUs, raster stream on all / the remaining CUs.
run() { BRIDGES_RASTER_STREAM="$1" timeout -k 10 120 python bench.py --no-cpu-baseline --no-other-modes 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-14s'%'$1', round(d['value']), round(d['ms_per_step'],4), 'raster ms', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3))" || echo "$1 failed"; }
for m in "${@:-}"; do run "$m"; done
