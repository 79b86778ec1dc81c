#!/bin/bash
# kernel durations of one SuccessorMLP optimiser step (rocprofv3 kernel trace of tools/mlp_step_bench.py); run through gpurun
set -o pipefail
R=$PWD
out=$R/gpurun_out/prof_mlp_step
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $R/tools/mlp_step_bench.py > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cd $R
f=$(find $out -name "*kernel_stats.csv" | head -1)
cp $f $out/kernel_stats.csv
head -12 $f | cut -c1-60,150-260
find $out -name "*kernel_trace.csv" -size +30M -delete
