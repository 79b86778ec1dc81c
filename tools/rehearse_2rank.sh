#!/bin/bash
# Rehearses bench.py's N>1 code path on ONE card: two ranks share cuda:0, reduction over gloo.
BENCH_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
    --master-port 29533 bench.py --gpus 2 --steps 30 --warmup 5 --envs 1024
