#!/bin/bash
set -o pipefail
out=gpurun_out/r3d
mkdir -p $out
for b in 0 1; do
  BRIDGES_SINGLE_ENV_BATCH=$b timeout -k 10 400 python tools/single_env_throughput.py --count_syncs --episodes 30 > $out/single_env_batch$b.json 2> $out/single_env_batch$b.err
  cat $out/single_env_batch$b.json
done
bash tools/profile_modes.sh r3d mlp
python tools/lockstep_kernels.py gpurun_out/prof_r3d_mlp --last 4 --top 70 > $out/mlp_lockstep_kernels.txt
cat $out/mlp_lockstep_kernels.txt | cut -c1-150
