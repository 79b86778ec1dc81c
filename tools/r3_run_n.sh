#!/bin/bash
# kernel list of one configs[2] training lock-step on the final round-3 tree (rocprofv3 kernel trace of tools/train_throughput.py)
set -o pipefail
R=$PWD
out=$R/gpurun_out/prof_r3n_mlp
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $R/tools/train_throughput.py --locksteps 12 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
cd $R
python tools/lockstep_kernels.py $out --last 4 --top 70 > $out/lockstep_kernels.txt
python tools/lockstep_kernels.py $out --last 4 --top 90 --width 260 > $out/lockstep_kernels_wide.txt
find $out -name "*kernel_trace.csv" -size +30M -delete
head -40 $out/lockstep_kernels.txt
