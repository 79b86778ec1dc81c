#!/bin/bash
# Two training ranks sharing ONE card (gloo): rehearses the all-gather of transition records and the replicated replay.
cd "$(dirname "$0")/.."
BRIDGES_DIST_BACKEND=gloo PYTHONPATH=bridges-with-reinforcement-learning_amd python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 \
    bridges-with-reinforcement-learning_amd/robotoddler/training/successor_dqn.py --model SuccessorMLP \
    --loss_function mse_q_values+mse_block_features --tower_height 4 --max_steps 15 --num_envs 256 --num_episodes 600 \
    --num_training_steps 2 --batch_size 16 --seed 3 --learning_rate 1e-4 --verbose
