#!/bin/bash
set -o pipefail
out=gpurun_out/r3e; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_api_golden.py tests/test_gpu_dqn.py tests/test_gpu_vec_dqn.py -m gpu -x -q > $out/tests.log 2>&1; rc=$?
tail -3 $out/tests.log
[ $rc -eq 0 ] || { tail -40 $out/tests.log; exit $rc; }
timeout -k 10 300 python tools/single_env_throughput.py --count_syncs --episodes 50 2>/dev/null | cut -c1-400
for i in 1 2; do
timeout -k 10 300 python tools/train_throughput.py --envs 4096 --tower 4 --max_steps 15 --model SuccessorMLP --loss mse_block_features --locksteps 12 --warmup 6 2>$out/train.err | grep "^{" > $out/train_$i.json; python -c "import json; d=json.load(open('$out/train_$i.json')); print({k: round(v,3) if isinstance(v,float) else v for k,v in d.items() if k not in ('config','note')})"
done
