#!/bin/bash
# Randomized parity record against the plain-C oracle (run on the box through gpurun): 5 tasks x 1024 envs x 300 lock-steps of
# the lock-step env, 3 tasks x 256 envs x 30 lock-steps of the fused candidate-stability mask.  usage: tools/stress_record.sh <out file> [seed]
out=${1:-gpurun_out/stress_record.txt}; seed=${2:-41}
mkdir -p $(dirname $out); : > $out
for t in tower4 hexbridge mixed bridge_mu05 tower2; do
  timeout -k 10 900 python tests/stress/stress_parity.py --envs 1024 --locksteps 300 --task $t --seed $seed 2>&1 | grep "RESULT\|MISMATCH" | tee -a $out
done
for t in tower4 hexbridge mixed; do
  timeout -k 10 900 python tests/stress/stress_candidate_stability.py --envs 256 --locksteps 30 --task $t --seed $((seed + 2)) 2>&1 | grep "RESULT\|MISMATCH" | tee -a $out
done
