#!/bin/bash
# HBM traffic of k_raster from the PMC counters, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: WRITE_SIZE and
# FETCH_SIZE in SEPARATE passes (they do not fit one pass), each with --kernel-trace only; FETCH_SIZE is doubled for
# gfx950 afterwards (tools/pmc_summarise.py).  Run on the box through gpurun from the repo root:  tools/pmc_k_raster.sh <tag>
set -o pipefail
tag=$1
root=$(pwd)
export TMPDIR=/tmp
for ctr in WRITE_SIZE FETCH_SIZE; do
  out=$root/gpurun_out/pmc_${tag}_$ctr
  (cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out -- python3 $root/bench.py --steps 20 --warmup 5 --groups 1 --no-cpu-baseline --no-other-modes --seeds 0) > $out.log 2>&1 || { tail -5 $out.log; exit 1; }
  grep "^{" $out.log | tail -1 > $out.json
done
python3 $root/tools/pmc_summarise.py $root/gpurun_out/pmc_${tag}_WRITE_SIZE $root/gpurun_out/pmc_${tag}_FETCH_SIZE > $root/gpurun_out/pmc_${tag}_k_raster.json
cat $root/gpurun_out/pmc_${tag}_k_raster.json
