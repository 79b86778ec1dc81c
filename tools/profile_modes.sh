#!/bin/bash
# rocprofv3 kernel stats of the bench modes (run on the MI355X box through gpurun, from the repo root):
#   tools/profile_modes.sh <tag> [sim] [cand] [hex] [mlp] [conv] [conv_all] [unet] [unet_all]     (*_all: every candidate row fed, --no_dedup)
# Writes gpurun_out/prof_<tag>_<mode>/ (trace + stats) and gpurun_out/prof_<tag>_<mode>.json (the bench line).
set -o pipefail
tag=$1; shift
root=$(pwd)
export TMPDIR=/tmp
for mode in "$@"; do
  out=$root/gpurun_out/prof_${tag}_${mode}
  case $mode in
    sim)  args="$root/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-modes" ;;
    cand) args="$root/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-modes --mode candidate-stability" ;;
    hex)  args="$root/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-other-modes --shapes hexagon --bridge_length 3" ;;
    mlp)  args="$root/tools/train_throughput.py --locksteps 6 --warmup 6 --envs 4096 --tower 4 --max_steps 15 --model SuccessorMLP --loss mse_block_features" ;;
    conv) args="$root/tools/train_throughput.py --locksteps 4 --warmup 12 --envs 1024 --tower 2 --max_steps 10 --model ConvNet --loss mse_q_values" ;;
    conv_all) args="$root/tools/train_throughput.py --locksteps 4 --warmup 12 --envs 1024 --tower 2 --max_steps 10 --model ConvNet --loss mse_q_values --no_dedup" ;;
    unet) args="$root/tools/train_throughput.py --locksteps 3 --warmup 12 --envs 4096 --max_steps 15 --model UNet --loss mse_q_values+mse_block_features --shapes hexagon --bridge_length 3" ;;
    unet_all) args="$root/tools/train_throughput.py --locksteps 2 --warmup 3 --envs 4096 --max_steps 15 --model UNet --loss mse_q_values+mse_block_features --shapes hexagon --bridge_length 3 --no_dedup" ;;
    *) echo "unknown mode $mode"; exit 2 ;;
  esac
  (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $args) > $out.log 2>&1 || { tail -20 $out.log; exit 1; }
  grep "^{" $out.log | tail -1 > $out.json
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $mode: $f"; head -12 $f | cut -c1-200
  # keep the summaries, drop the raw trace (large)
  true
done
