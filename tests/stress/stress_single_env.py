"""Randomized run of the single-environment drop-in surface (assembly_gym.AssemblyGym + robotoddler.utils.actions + the feature
functions of successor_dqn.py) against the numpy + HiGHS oracle (oracle/env.py OracleGym), episode by episode with a random
policy over the FILTERED actions: candidate list, filter mask, candidate rasters and linear rewards, then the step's stable flag,
reward, termination / truncation, stabilities_freezing(), the state raster and distance_to_targets.
    python tests/stress/stress_single_env.py --episodes 60 --task mixed --seed 5"""
import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--episodes", type=int, default=40)
ap.add_argument("--task", default="tower2")
ap.add_argument("--seed", type=int, default=1)
a = ap.parse_args()

from assembly_gym.envs.assembly_env import AssemblyEnv
from assembly_gym.envs import gym_env as G
from assembly_gym.utils.rendering import render_blocks_2d
from oracle import env as O
from robotoddler.training.successor_dqn import get_action_features, get_state_features, get_task_features
from robotoddler.utils.actions import filter_actions, generate_actions

TASKS = dict(tower2=("bridge_setup", dict(num_stories=2), 10, 0.8), tower4=("bridge_setup", dict(num_stories=4), 15, 0.8),
             hexbridge=("horizontal_bridge_setup", dict(num_obstacles=3, trapezoid=False, hexagon=True), 15, 0.8),
             mixed=("horizontal_bridge_setup", dict(num_obstacles=4, trapezoid=True, hexagon=True), 12, 2.0),
             bridge_mu05=("horizontal_bridge_setup", dict(num_obstacles=5), 15, 0.5))
name, kw, max_steps, mu = TASKS[a.task]
dev = torch.device("cuda")
xlim, ylim = O.XLIM, O.YLIM
env = G.AssemblyGym(**getattr(G, name)(**kw), reward_fct=G.sparse_reward, restrict_2d=True, max_steps=max_steps,
                    assembly_env=AssemblyEnv(render=False, mu=mu))
og = O.OracleGym(**getattr(O, name)(**kw), max_steps=max_steps, mu=mu)
rng = random.Random(a.seed)
t0, steps, mism = time.time(), 0, 0


def bad(what, *info):
    global mism
    mism += 1
    print("MISMATCH", what, *info, flush=True)
    if mism > 10:
        sys.exit(1)


for ep in range(a.episodes):
    obs, _ = env.reset()
    og.reset()
    reward_f, obstacle_f = get_task_features(obs, xlim=xlim, ylim=ylim, img_size=(64, 64), device=dev)
    if not np.array_equal(obstacle_f[0].cpu().numpy() > 0, og.obstacle_raster):
        bad("obstacle raster", ep)
    if not np.allclose(reward_f[0].cpu().numpy(), og.reward_map, rtol=1e-5, atol=1e-6):
        bad("reward map", ep)
    done = False
    while not done:
        cand = og.candidates()
        acts = [*generate_actions(env, x_discr_ground=O.X_DISCR_GROUND, offset_values=O.OFFSET_VALUES)]
        desc = [(x.target_block, x.target_face, x.shape, x.face, float(x.offset_x), float(x.offset_y)) for x in acts]
        if desc != [tuple(c) for c in cand["actions"]]:
            bad("candidate list", ep, len(desc), len(cand["actions"]))
            break
        block_f, _bin = get_state_features(obs, xlim=xlim, ylim=ylim, img_size=(64, 64), device=dev)
        if not np.array_equal(block_f[0].cpu().numpy() > 0, cand["state"]):
            bad("state raster", ep)
        feats = get_action_features(env, acts, xlim=xlim, ylim=ylim, img_size=(64, 64), device=dev)
        if not np.array_equal(feats[:, 0].cpu().numpy() > 0, cand["rasters"]):
            bad("candidate rasters", ep)
        kept, kept_f = filter_actions(env, acts, feats, block_features=block_f, obstacle_features=obstacle_f, xlim=xlim, ylim=ylim)
        keep_idx = [i for i, x in enumerate(acts) if any(x is k for k in kept)]
        if keep_idx != list(np.flatnonzero(cand["mask"])):
            bad("filter mask", ep, keep_idx[:8], list(np.flatnonzero(cand["mask"]))[:8])
            break
        lin = (kept_f[:, 0] * reward_f[0]).sum(dim=(-1, -2)).cpu().numpy()
        if not np.allclose(lin, cand["lin_reward"][keep_idx], rtol=1e-5, atol=1e-6):
            bad("linear reward", ep)
        if not keep_idx:
            break
        k = rng.randrange(len(keep_idx))
        act = acts[keep_idx[k]]
        obs, r, term, trunc, _ = env.step(act)
        stable, r2, term2, trunc2 = og.step(cand["actions"][keep_idx[k]])
        steps += 1
        if (obs["stable"], r, bool(term), bool(trunc)) != (stable, r2, term2, trunc2):
            bad("step", ep, (obs["stable"], r, bool(term), bool(trunc)), (stable, r2, term2, trunc2))
        if env.stabilities_freezing() != og.stabilities_freezing():
            bad("stabilities_freezing", ep, env.stabilities_freezing(), og.stabilities_freezing())
        img = render_blocks_2d(obs["blocks"], xlim=xlim, ylim=ylim, img_size=(64, 64))
        if not np.array_equal(img, og.state_raster()):
            bad("state raster after step", ep)
        if not np.allclose(obs["distance_to_targets"], og.distance_to_targets(), rtol=0, atol=1e-12):
            bad("distance_to_targets", ep, obs["distance_to_targets"], og.distance_to_targets())
        done = bool(term or trunc)
    if ep % 10 == 9:
        print(f"episode {ep + 1}: {steps} env-steps, {mism} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"RESULT task={a.task} episodes={a.episodes} seed={a.seed}: {steps} env-steps through the drop-in surface against the numpy oracle, {mism} mismatches")
sys.exit(1 if mism else 0)
