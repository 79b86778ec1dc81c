#!/bin/bash
# Does the rasteriser starve the other env groups' task kernels?  k_raster's 8 waves per SIMD (64 VGPRs each) fill the register
# file; a k_step launched at the same moment waits for slots (profiles/r04_kstep_tail.txt).  This builds variants whose rasteriser
# workgroups carry unused dynamic LDS (RASTER_DYN_LDS bytes: 160 KB / that = workgroups of 4 waves per CU) and runs the headline
# bench with each.   tools/raster_occupancy.sh <out dir> [bytes ...]
set -e
cd "$(dirname "$0")/.."
OUT=${1:-gpurun_out/raster_occ}; shift || true
SIZES=${@:-"0 20480 26624 32768"}
mkdir -p "$OUT"
HASH=$(PYTHONPATH=bridges-with-reinforcement-learning_amd python3 -c "from bridges_hip import abi; print(abi.source_hash())")
for b in $SIZES; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared -std=c++17 "-DBRIDGES_SRC_HASH=\"$HASH\"" -DRASTER_DYN_LDS=$b \
      bridges-with-reinforcement-learning_amd/csrc/api.hip -o tools/libbridges_hip_cs_lds$b.so
  for g in 2 3; do
    BRIDGES_LIB=tools/libbridges_hip_cs_lds$b.so python3 bench.py --no-cpu-baseline --no-other-modes --groups $g --steps 200 --warmup 30 > "$OUT/lds$b.g$g.json" 2> "$OUT/lds$b.g$g.err" || echo "bench failed"
    python3 -c "
import json,sys
j=json.loads(open('$OUT/lds$b.g$g.json').read().strip().splitlines()[-1])
print('dyn LDS $b groups $g: value %.4g  per seed %s  raster launch %.3f ms  frac %.3f' % (j['value'], [round(p['value']/1e6,3) for p in j['config']['per_seed']], j['roofline']['avg_launch_ms'], j['roofline']['frac']))"
  done
done
