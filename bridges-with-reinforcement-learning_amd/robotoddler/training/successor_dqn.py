"""Successor-feature DQN with the reference's CLI surface (robotoddler/training/successor_dqn.py of the reference).

Two ways to run:

* ``--num_envs 1`` (default): the reference's single-environment loop -- rollout_episode / train_policy_net /
  update_target_net with the same signatures and Transition layout -- on the assembly_gym drop-in (every
  placement, stability solve and raster is a HIP operator call).
* ``--num_envs N`` (N > 1): N environments in lock-step on the GPU (VecAssemblyGym) with a device replay buffer
  of compact transition records that are re-rasterised when sampled (robotoddler/training/vec_dqn.py).

Flags are the reference's (successor_dqn.py:573-596) plus ``--tower_height`` (README / BASELINE.json name for
``bridge_setup(num_stories=N)``; absent from the reference's argparse at HEAD), ``--num_envs``, ``--shapes``.
There is no CPU path: ``--device cpu`` is rejected.
"""
import argparse
import os
import random
import sys
from collections import namedtuple

import numpy as np
import torch

_PKG = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from assembly_gym.envs.assembly_env import AssemblyEnv, Block, Shape                      # noqa: E402
from assembly_gym.envs.gym_env import (AssemblyGym, bridge_setup, horizontal_bridge_setup,  # noqa: E402
                                       sparse_reward)
from assembly_gym.utils.rendering import render_blocks_2d_bits                            # noqa: E402
from bridges_hip import dqn_ops, ops                                                      # noqa: E402
from robotoddler.models.cv import ConvNet, Policy, SuccessorMLP                           # noqa: E402
from robotoddler.utils.actions import filter_actions, generate_actions                    # noqa: E402
from robotoddler.utils.replay_memory import PrioritizedReplayBuffer, ReplayBuffer          # noqa: E402
from robotoddler.utils.utils import (convolve_with_gaussian, init_weights, parse_img_size,   # noqa: E402
                                     save_checkpoint)

Transition = namedtuple('Transition',
                        ('block_features', 'binary_features', 'action', 'action_features', 'reward', 'lin_reward',
                         'done', 'reward_features', 'obstacle_features', 'next_block_features',
                         'next_binary_features', 'next_available_actions', 'next_actions_features',
                         'next_reward_features', 'next_obstacle_features', 'td_error'))


def _image(bits, img_size):
    """bit raster(s) -> f32 [n, S, S] (contiguous)."""
    return ops.crop(ops.bits_to_f32(bits), img_size).contiguous()


# ---- feature extraction (successor_dqn.py:47-94) ---------------------------------------------------------------
def _state_features(observation, xlim, ylim, img_size, device):
    """get_state_features plus the state's bit raster (what the candidate filter tests overlaps against)."""
    binary = [observation['stable'], observation['collision'], observation['collision_block'],
              observation['collision_obstacle'], observation['collision_floor'], observation['collision_boundary']]
    bits = render_blocks_2d_bits(observation['blocks'], xlim, ylim, img_size)
    image = _image(bits, img_size)                                                                     # [1,S,S] f32
    binary_t, = ops.upload(image.device, np.asarray(binary, dtype=np.float32))                         # (no blocking copy)
    return image.to(device), binary_t.to(device), bits


def get_state_features(observation, xlim=(0, 1), ylim=(0, 1), img_size=(64, 64), device=None):
    image, binary, _bits = _state_features(observation, xlim, ylim, img_size, device)
    return image, binary


def get_task_features(obs, xlim=(0, 1), ylim=(0, 1), img_size=(64, 64), device=None):
    cube = Shape(urdf_file='shapes/cube06.urdf')
    target_blocks = [Block(shape=cube, position=target) for target in obs['targets']]
    reward = _image(render_blocks_2d_bits(target_blocks, xlim, ylim, img_size), img_size)[0]
    reward = convolve_with_gaussian(reward, 101, 16)                                      # successor_dqn.py:80-82
    obstacle = _image(render_blocks_2d_bits(obs['obstacle_blocks'], xlim, ylim, img_size), img_size)
    return reward.unsqueeze(0).to(device), obstacle.to(device)


def get_action_features(env, actions, xlim=(0, 1), ylim=(0, 1), img_size=(64, 64), device=None):
    # one batched bridges_create_block call for all candidates when the gym offers it (the reference: one call each)
    blocks = env.create_blocks(actions) if hasattr(env, "create_blocks") else [env.create_block(action) for action in actions]
    return _image(ops.raster_bits(blocks, xlim, ylim, img_size), img_size).unsqueeze(1).to(device)   # [A,1,S,S]


# ---- policies (successor_dqn.py:98-132) -------------------------------------------------------------------------
class EpsilonGreedy:
    """epsilon-greedy with count-based exploration: explore = the action whose raster overlaps least with the
    rasters already tried at this episode step."""

    def __init__(self, eps_start=0.5, eps_end=0.05, gamma=0.99, episode=0, max_steps=10, device=torch.device('cpu')):
        self.epsilon = (eps_start - eps_end) * (gamma ** episode) + eps_end
        self.eps_start, self.eps_end, self.gamma = eps_start, eps_end, gamma
        self.max_steps, self.device = max_steps, device
        self.step_images = [torch.zeros(64, 64, device=device) for _ in range(max_steps)]

    def step(self):
        self.epsilon = (self.epsilon - self.eps_end) * self.gamma + self.eps_end
        return self

    def __call__(self, q_values, step_index, action_features, *args, **kwargs):
        if random.random() > self.epsilon:
            return torch.argmax(q_values).item()
        if step_index >= len(self.step_images):            # the reference indexes past max_steps=10 (latent IndexError)
            self.step_images += [torch.zeros(64, 64, device=self.device) for _ in range(step_index + 1 - len(self.step_images))]
        feats = action_features.squeeze(1).to(self.device)
        if self.step_images[step_index].shape != feats.shape[-2:]:       # --image_size other than the 64x64 default
            self.step_images[step_index] = torch.zeros(feats.shape[-2:], device=self.device)
        join = torch.sum(self.step_images[step_index] * feats, dim=(-1, -2))
        sel = torch.argmin(join).item()
        self.step_images[step_index] += feats[sel]
        return sel


# ---- training (successor_dqn.py:157-288) ------------------------------------------------------------------------
def _task_key(*tensors):
    """Fingerprint of an episode's task features (reward map, obstacle raster): transitions whose tensors carry the same key
    share them VALUE for value, which is what lets a batch go through the hand-written optimiser step (one reward / obstacle
    vector per batch).  One host read per episode."""
    import hashlib
    h = hashlib.sha1()
    for t in tensors:
        h.update(t.detach().float().cpu().numpy().tobytes())
    return h.hexdigest()


def _tagged(t, key):
    t._task_key = key
    return t


class _FusedTrainer:
    """train_policy_net's optimiser steps for a SuccessorMLP on the hand-written step of the vectorised loop
    (bridges_hip/mlp_ops.py FusedSuccessorStep: forward, both MSE losses, backward and Adam as ~11 launches on the f32
    matrix cores instead of ~60 library / element-wise ones, no host wait per step).  Same losses as the autograd form:
    the reference's q target is the [B, B] broadcast  lin_reward[j] + gamma q'[i]  (successor_dqn.py:222 with :435), whose
    mean-squared error against q[i] is  mean_i (q[i] - (mean(lin) + gamma q'[i]))^2 + var(lin)  -- the same gradient as the
    element-wise target mean(lin) + gamma q'[i]; the constant var(lin) is added to the logged loss."""

    def __init__(self, policy_net, optimizer, batch_size, loss_parts):
        from bridges_hip.mlp_ops import FusedSuccessorStep
        self.key = (id(optimizer), batch_size, tuple(loss_parts))
        self.use_q, self.use_sf = 'mse_q_values' in loss_parts, 'mse_block_features' in loss_parts
        self.step = FusedSuccessorStep(policy_net, batch_size, self.use_q, self.use_sf, optimizer=optimizer)
        if not self.step.fused_adam:
            raise ValueError("the optimiser is not a plain Adam over the net's flattened parameters")
        dev = self.step.flat.device
        self.counter = torch.zeros((), dtype=torch.int64, device=dev)
        self.loss1 = torch.zeros(1, dtype=torch.float32, device=dev)
        self._graphs, self._seen = {}, {}

    def _steps(self, n, *tensors):
        """n optimiser steps on the n batches in ``tensors`` (block, action, binary, reward, obstacle, q target, sf target,
        losses).  The first call with a given n queues n x 11 launches; from the second on the sequence is one HIP graph over
        static copies of the inputs (a launch costs the host ~7 us, a step is ~100 us of GPU work: queued one by one the host
        and the GPU take about the same time, as a graph the host is free after one launch).  BRIDGES_TRAIN_GRAPH=0: never."""
        st = self.step
        self.counter.zero_()
        g = self._graphs.get(n)
        if g is None:
            seen = self._seen[n] = self._seen.get(n, 0) + 1
            if seen == 1 or os.environ.get("BRIDGES_TRAIN_GRAPH", "1") != "1":
                for _ in range(n):
                    st.launch(self.counter, *tensors)
                return tensors[-1]
            bufs = [torch.empty_like(t) if t is not None else None for t in tensors]
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(n):
                    st.launch(self.counter, *bufs)
            g = self._graphs[n] = (graph, bufs)
        graph, bufs = g
        for dst, src in zip(bufs[:-1], tensors[:-1]):
            if dst is not None:
                dst.copy_(src)
        bufs[-1].zero_()
        graph.replay()
        return bufs[-1].clone()

    @staticmethod
    def applies(policy_net, optimizer, loss_parts, scheduler, transitions, batch=None, device=None):
        if scheduler is not None or not isinstance(policy_net, SuccessorMLP) or getattr(policy_net, "_flat_params", None) is None:
            return False
        if not set(loss_parts) <= {'mse_q_values', 'mse_block_features'} or type(optimizer) is not torch.optim.Adam:
            return False
        if not transitions:
            return False
        if batch is not None:
            if not batch.block_features.is_cuda or batch.block_features.dtype != torch.float32:
                return False
        elif torch.device(device).type != 'cuda' or any(t.block_features.dtype != torch.float32 for t in transitions):
            return False
        keys = {getattr(t.reward_features, "_task_key", None) for t in transitions}
        return len(keys) == 1 and None not in keys and all(getattr(t.obstacle_features, "_task_key", None) in keys for t in transitions)

    @classmethod
    def of(cls, policy_net, optimizer, batch_size, loss_parts):
        """The net's trainer for this optimiser / batch size / loss, built on first use; None if the optimiser is not a plain
        Adam over the flattened parameters."""
        tr = getattr(policy_net, "_fused_trainer", None)
        if tr is None or tr.key != (id(optimizer), batch_size, tuple(loss_parts)):
            try:
                sync_fused_optimizer(policy_net)
                tr = policy_net._fused_trainer = cls(policy_net, optimizer, batch_size, loss_parts)
            except ValueError:
                return None
        return tr

    def run_all(self, drawn, target_net, gamma, device):
        """All optimiser steps of one train_policy_net call: ``drawn`` = the sampled transitions of every step.  The target net
        does not change inside the call, so the TD targets of all steps come from ONE target forward and ONE segmented argmax
        over all next-action rows, and the steps are launch sequences that read their batch by a device-side counter
        (FusedSuccessorStep.launch): per step the host queues 11 launches and nothing else.  A transition drawn several times
        (the draws are with replacement across the steps; 20 x 32 draws from a buffer of a few hundred) is stacked and
        evaluated once and its rows are gathered; the rows of a next state are gathered from one copy per transition; the
        task's reward / obstacle rasters are the same image for every row."""
        n, st = len(drawn), self.step
        B, px = st.batch, st.px
        slot, uniq, which = {}, [], []
        for b in drawn:
            for t in b:
                k = slot.get(id(t))
                if k is None:
                    k = slot[id(t)] = len(uniq)
                    uniq.append(t)
                which.append(k)
        cat = lambda field: torch.cat([getattr(t, field) for t in uniq]).to(device=device)
        num_actions = [max(1, len(t.next_available_actions)) for t in uniq]
        assert [t.next_actions_features.shape[0] for t in uniq] == num_actions, "a next state's action rows and its action list differ"
        seg_np = np.zeros(len(uniq) + 1, dtype=np.int32)
        np.cumsum(num_actions, out=seg_np[1:])
        owner = np.repeat(np.arange(len(uniq), dtype=np.int64), num_actions)
        seg, done, owner, which = ops.upload(device, seg_np, np.asarray([t.done for t in uniq], dtype=np.bool_), owner,
                                             np.asarray(which, dtype=np.int64))
        rows = int(seg_np[-1])
        reward, obstacle = uniq[0].reward_features.to(device), uniq[0].obstacle_features.to(device)
        with torch.no_grad():
            action_u = cat('action_features')
            one_row = lambda x: x.shape[0] == 1 or x.stride(0) == 0              # rollout_episode stores expand()ed views
            if all(one_row(t.next_block_features) and one_row(t.next_binary_features) for t in uniq):
                nb = torch.cat([t.next_block_features[:1] for t in uniq]).to(device).index_select(0, owner)
                nbin = torch.cat([t.next_binary_features[:1] for t in uniq]).to(device).index_select(0, owner)
            else:
                nb, nbin = cat('next_block_features'), cat('next_binary_features')
            next_q, next_sf, _ = target_net(nb, nbin, cat('next_actions_features'), reward.expand(rows, -1, -1, -1),
                                            obstacle.expand(rows, -1, -1, -1))
            zeros = torch.zeros(len(uniq), dtype=torch.float32, device=device)
            nq = next_q.contiguous().float()
            q_sel, _, _ = dqn_ops.td_target(seg, nq, zeros, done, 1.0)
            sf_target = None
            if self.use_sf:
                _, sf_target, _ = dqn_ops.td_target(seg, nq, zeros, done, gamma, next_sf=next_sf[:, 0], action_raster=action_u.squeeze(1))
                sf_target = sf_target.reshape(len(uniq), px).index_select(0, which)
            st.check_hyperparameters()
            q_target = extra = None
            if self.use_q:
                lin = cat('lin_reward').reshape(-1).float().index_select(0, which).view(n, B)
                m = lin.mean(dim=1, keepdim=True)
                q_target = (m + gamma * q_sel.index_select(0, which).view(n, B)).reshape(-1).contiguous()
                extra = ((lin - m) ** 2).mean(dim=1)
            losses = torch.zeros(n, dtype=torch.float32, device=device)
            losses = self._steps(n, cat('block_features').reshape(len(uniq), px).index_select(0, which),
                                 action_u.reshape(len(uniq), px).index_select(0, which), cat('binary_features').index_select(0, which),
                                 reward.reshape(px).contiguous(), obstacle.reshape(px).contiguous(), q_target, sf_target, losses)
        return losses + extra if extra is not None else losses

    def run(self, batch, q_sel, sf_target, gamma):
        B = batch.block_features.shape[0]
        st = self.step
        st.check_hyperparameters()
        lin = batch.lin_reward.reshape(-1).float()
        extra = None
        q_target = None
        if self.use_q:
            m = lin.mean()
            q_target = (m + gamma * q_sel).contiguous()
            extra = ((lin - m) ** 2).mean()
        self.counter.zero_()
        px = st.px
        st.launch(self.counter, batch.block_features.reshape(B, px).contiguous(), batch.action_features.reshape(B, px).contiguous(),
                  batch.binary_features.contiguous(), batch.reward_features[0].reshape(px).contiguous(),
                  batch.obstacle_features[0].reshape(px).contiguous(), q_target,
                  sf_target.reshape(B, px).contiguous() if self.use_sf else None, self.loss1)
        return self.loss1[0] + extra if extra is not None else self.loss1[0].clone()


def sync_fused_optimizer(policy_net):
    """Hand Adam's step count back to the torch optimiser (its moments already are the fused step's buffers): call before
    optimizer.state_dict() / optimizer.step() when train_policy_net may have run on the hand-written step."""
    tr = getattr(policy_net, "_fused_trainer", None)
    if tr is not None:
        tr.step.export_state()


def train_policy_net(policy_net, target_net, optimizer, replay_buffer, gamma, loss_fct='mse_q_values', scheduler=None,
                     n_steps=10, batch_size=16, verbose=False, device='cuda'):
    if len(replay_buffer) < batch_size:
        return
    loss_fct = loss_fct.split('+')
    policy_net.train()
    target_net.eval()
    mse = torch.nn.MSELoss()
    losses = []
    # the draws of all steps first: they depend on the buffer and the random generators only, which the steps do not touch
    # (a buffer without draw(): one sample() per step, as the reference)
    draw = getattr(replay_buffer, "draw", None)
    if draw is not None:
        from robotoddler.utils.replay_memory import _stack
        drawn = [draw(batch_size) for _ in range(n_steps)]
        if batch_size is not None and _FusedTrainer.applies(policy_net, optimizer, loss_fct, scheduler, [t for b in drawn for t in b], device=device):
            tr = _FusedTrainer.of(policy_net, optimizer, batch_size, loss_fct)
            if tr is not None:
                out = []
                for k in range(0, n_steps, 64):           # (bounds the target pass: 64 steps x 32 transitions x their next actions)
                    out.append(tr.run_all(drawn[k:k + 64], target_net, gamma, device))
                return torch.cat(out).tolist()            # ONE host read for all steps
        steps = ((transitions, _stack(transitions, device)) for transitions in drawn)
    else:
        steps = (replay_buffer.sample(batch_size=batch_size, stack_tensors=True, device=device) for _ in range(n_steps))
    for transitions, batch in steps:
        fused = _FusedTrainer.applies(policy_net, optimizer, loss_fct, scheduler, transitions, batch)
        if fused:
            tr = _FusedTrainer.of(policy_net, optimizer, batch.block_features.shape[0], loss_fct)
            fused = tr is not None
        if not fused and getattr(policy_net, "_fused_trainer", None) is not None:
            sync_fused_optimizer(policy_net)                  # optimizer.step() takes over: it needs the true step count,
            policy_net._fused_trainer = None                  # and a later fused batch adopts the optimiser's state afresh
        if fused:
            with torch.no_grad():
                next_q, next_sf, _next_bin = target_net(
                    batch.next_block_features, batch.next_binary_features, batch.next_actions_features,
                    batch.next_reward_features, batch.next_obstacle_features)
                num_actions = [max(1, len(a)) for a in batch.next_available_actions]
                seg, done = ops.upload(next_q.device, np.cumsum([0] + num_actions).astype(np.int32), np.asarray(batch.done, dtype=np.bool_))
                zeros = torch.zeros(len(num_actions), dtype=torch.float32, device=next_q.device)
                nq = next_q.contiguous().float()
                q_sel, _, _ = dqn_ops.td_target(seg, nq, zeros, done, 1.0)
                sf_target = None
                if 'mse_block_features' in loss_fct:
                    _, sf_target, _ = dqn_ops.td_target(seg, nq, zeros, done, gamma, next_sf=next_sf[:, 0],
                                                        action_raster=batch.action_features.squeeze(1))
            losses.append(tr.run(batch, q_sel, sf_target, gamma))
            continue
        q_values, succ_block_features, succ_binary_features = policy_net(
            batch.block_features, batch.binary_features, batch.action_features, batch.reward_features,
            batch.obstacle_features)
        with torch.no_grad():
            next_q, next_sf, _next_bin = target_net(
                batch.next_block_features, batch.next_binary_features, batch.next_actions_features,
                batch.next_reward_features, batch.next_obstacle_features)
            num_actions = [max(1, len(a)) for a in batch.next_available_actions]
            seg = torch.tensor(np.cumsum([0] + num_actions), dtype=torch.int32, device=next_q.device)
            done = torch.tensor(batch.done, dtype=torch.bool, device=next_q.device)
            use_sf = 'mse_block_features' in loss_fct
            if use_sf and succ_block_features is None:
                raise ValueError("No successor block features available from the chosen policy net.")
            # fused HIP op: segmented argmax over the ragged next-action rows, done masking, a + gamma psi'.
            # lin_reward = 0, gamma_q = 1 makes q_sel the done-masked next q of the selected action; the reward is
            # added below with torch broadcasting, because the reference adds a [B,1] lin_reward to a [B] vector
            # (successor_dqn.py:222 with :435) and thereby trains on a [B,B] target -- reproduced as is.
            zeros = torch.zeros(len(num_actions), dtype=torch.float32, device=next_q.device)
            q_sel, _, sel_rows = dqn_ops.td_target(seg, next_q.contiguous().float(), zeros, done, 1.0)
            sf_target = None
            if use_sf:
                _, sf_target, _ = dqn_ops.td_target(seg, next_q.contiguous().float(), zeros, done, gamma,
                                                    next_sf=next_sf[:, 0],
                                                    action_raster=batch.action_features.squeeze(1))
        loss = 0.
        if 'mse_q_values' in loss_fct:
            loss = loss + mse(q_values, batch.lin_reward + gamma * q_sel)
        if use_sf:
            loss = loss + mse(succ_block_features[:, 0], sf_target.view_as(succ_block_features[:, 0]))
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        if scheduler is not None:
            scheduler.step()
        losses.append(loss.detach())
    return torch.stack([l.reshape(()) for l in losses]).tolist() if losses else []      # ONE host read for all steps


def update_target_net(policy_net, target_net, tau=0.01):
    """theta_target <- tau * theta + (1 - tau) * theta_target (successor_dqn.py:280-288) with the HIP soft-update
    kernel: one launch when both nets were flattened (dqn_ops.FlatParameters), else one per tensor."""
    fp, ft = getattr(policy_net, "_flat_params", None), getattr(target_net, "_flat_params", None)
    if fp is not None and ft is not None:
        dqn_ops.soft_update_(ft.flat, fp.flat, tau)
        return
    tsd, psd = target_net.state_dict(), policy_net.state_dict()
    for key, p in psd.items():
        t = tsd[key]
        if t.dtype == torch.float32 and t.is_contiguous() and p.is_contiguous() and t.data_ptr() % 16 == 0 and p.data_ptr() % 16 == 0:
            dqn_ops.soft_update_(t, p, tau)
        else:                                             # non-float / unaligned buffers: not on the hot path
            t.copy_(p * tau + t * (1 - tau))


def flatten_nets(*nets):
    for n in nets:
        n._flat_params = dqn_ops.FlatParameters(n)


# ---- rollout (successor_dqn.py:365-475) -------------------------------------------------------------------------
def rollout_episode(env, policy, policy_net, x_discr_ground, setup_fct, offset_values=[0.], img_size=(64, 64),
                    xlim=(0, 1), ylim=(0, 1), log_images=False, device=None):
    done, transitions = False, []
    images = [] if log_images else None
    policy_net.eval()
    obs, info = env.reset(**setup_fct())
    kw = dict(img_size=img_size, device=device, xlim=xlim, ylim=ylim)
    reward_features, obstacle_features = get_task_features(obs, **kw)
    task_key = _task_key(reward_features, obstacle_features)        # lets train_policy_net batch transitions of one task
    block_features, binary_features, state_bits = _state_features(obs, xlim, ylim, img_size, device)
    obstacle_bits = render_blocks_2d_bits(obs['obstacle_blocks'], xlim, ylim, img_size)
    batched = hasattr(env, "create_blocks") and tuple(img_size)[0] == tuple(img_size)[1] <= 64

    def candidates(block_f, bits_s):
        """generate_actions -> get_action_features -> filter_actions (successor_dqn.py:375-377) of the current state.  With the
        drop-in gym the three are ONE operator call (bridges_action_features: rasters of every candidate, the bounds test of
        collision_on_action and the two overlap tests of filter_actions; tests/test_gpu_api_golden.py shows it equal to the
        composition) and one mask copied back, instead of a reduction pair and a host read per action."""
        acts = [*generate_actions(env, x_discr_ground=x_discr_ground, offset_values=offset_values)]
        if not batched or not acts:
            feats = get_action_features(env, acts, **kw)
            return filter_actions(env, acts, feats, block_features=block_f, obstacle_features=obstacle_features, xlim=xlim, ylim=ylim)
        _bits, img, mask, _lin = ops.action_features(env.create_blocks(acts), xlim, ylim, state_bits=bits_s, obstacle_bits=obstacle_bits,
                                                     img_size=img_size, want_f32=True)
        keep = np.flatnonzero(mask.cpu().numpy())                                          # the one host read of the filter
        rows, = ops.upload(img.device, keep.astype(np.int64))
        feats = ops.crop(img, img_size).index_select(0, rows).unsqueeze(1).contiguous().to(device)
        return [acts[i] for i in keep], feats

    available_actions, action_features = candidates(block_features, state_bits)
    num_actions = len(available_actions)
    step_index = 0

    def qnet(bf, binf, af, n):
        with torch.no_grad():
            return policy_net(bf.expand(n, -1, -1, -1), binf.expand(n, -1), af, reward_features.expand(n, -1, -1, -1),
                              obstacle_features.expand(n, -1, -1, -1))

    q_next = None
    while not done:
        # (the net's outputs for this state were computed as "next state" of the previous step: the same net in eval mode on
        # the same rows -- the reference runs the forward a second time, successor_dqn.py:383 after :420)
        q_values, succ_block_features, succ_binary_features = q_next if q_next is not None else qnet(
            block_features, binary_features, action_features, num_actions)
        sel = policy(q_values, step_index, action_features, succ_block_features, succ_binary_features)
        sel = int(sel)
        action = available_actions[sel]
        selected_action_features = action_features[sel]
        next_observation, reward, terminated, truncated, info = env.step(action)
        done = bool(terminated or truncated)
        frozen_stable, unfrozen_stable = env.stabilities_freezing()
        lin_reward = torch.zeros(1, device=device).view(-1)
        if frozen_stable:
            lin_reward = torch.sum(selected_action_features * reward_features).view(-1) / 100
        if unfrozen_stable:
            lin_reward = torch.sum(selected_action_features * reward_features).view(-1)
        next_block_features, next_binary_features, state_bits = _state_features(next_observation, xlim, ylim, img_size, device)
        next_available_actions, next_action_features = candidates(next_block_features, state_bits)
        num_actions = len(next_available_actions)
        if num_actions == 0:
            done = True
            next_action_features = torch.zeros([1, 1, *img_size], device=device)
        # q(s, a) and max_a' q(s', a') travel to the host together (one read instead of two)
        pair = [q_values[sel].reshape(1).float()]
        q_next = None
        if not done:
            q_next = qnet(next_block_features, next_binary_features, next_action_features, num_actions)
            pair.append(q_next[0].max().reshape(1).float())
        pair = torch.cat(pair).tolist()
        q_value, next_q_value = pair[0], (pair[1] if not done else 0)
        td_error = abs(q_value - (reward + 0.95 * next_q_value))          # successor_dqn.py:425 (hard-coded 0.95)
        n1 = max(1, num_actions)
        transitions.append(Transition(
            block_features=block_features.unsqueeze(0), binary_features=binary_features.unsqueeze(0),
            action_features=selected_action_features.unsqueeze(0), reward_features=_tagged(reward_features.unsqueeze(0), task_key),
            obstacle_features=_tagged(obstacle_features.unsqueeze(0), task_key), action=action, lin_reward=lin_reward.unsqueeze(0),
            reward=torch.Tensor([reward]), done=done,
            next_block_features=next_block_features.expand(n1, -1, -1, -1),
            next_binary_features=next_binary_features.expand(n1, -1), next_actions_features=next_action_features,
            next_reward_features=reward_features.expand(n1, -1, -1, -1),
            next_obstacle_features=obstacle_features.expand(n1, -1, -1, -1),
            next_available_actions=next_available_actions, td_error=td_error))
        block_features, binary_features = next_block_features, next_binary_features
        action_features, available_actions = next_action_features, next_available_actions
        if log_images:
            if succ_block_features is None:
                raise ValueError("No successor block features available from the chosen policy net. Disable image "
                                 "logging or use a different model.")
            images.append(dict(succ_block_features=succ_block_features[sel][0].cpu().numpy()))
        step_index += 1
    return transitions, images


def rollout_episode_scripted(env, predefined_actions, setup_fct, x_discr_ground, offset_values=[0.], img_size=(64, 64),
                             xlim=(-3, 7), ylim=(0., 10), log_images=False, device=None):
    """Roll out predefined actions (successor_dqn.py:290-362): lin_reward = sum(action raster * reward map) without
    the stability scaling, td_error = 0."""
    done, transitions, images = False, [], ([] if log_images else None)
    obs, info = env.reset(**setup_fct())
    kw = dict(img_size=img_size, device=device, xlim=xlim, ylim=ylim)
    reward_features, obstacle_features = get_task_features(obs, **kw)
    block_features, binary_features = get_state_features(obs, **kw)
    for action in predefined_actions:
        if done:
            break
        selected_action_features = get_action_features(env, [action], **kw)[0]
        next_observation, reward, terminated, truncated, info = env.step(action)
        done = bool(terminated or truncated)
        lin_reward = torch.sum(selected_action_features * reward_features)
        next_block_features, next_binary_features = get_state_features(next_observation, **kw)
        acts = [*generate_actions(env, x_discr_ground=x_discr_ground, offset_values=offset_values)]
        feats = get_action_features(env, acts, **kw)
        next_available_actions, next_action_features = filter_actions(env, acts, feats, next_block_features,
                                                                      obstacle_features=obstacle_features, xlim=xlim, ylim=ylim)
        n = len(next_available_actions)
        transitions.append(Transition(
            block_features=block_features.unsqueeze(0), binary_features=binary_features.unsqueeze(0),
            action_features=selected_action_features.unsqueeze(0), reward_features=reward_features.unsqueeze(0),
            obstacle_features=obstacle_features.unsqueeze(0), action=action, lin_reward=lin_reward.unsqueeze(0),
            reward=torch.Tensor([reward]), done=done, next_block_features=next_block_features.expand(n, -1, -1, -1),
            next_binary_features=next_binary_features.expand(n, -1), next_actions_features=next_action_features,
            next_reward_features=reward_features.expand(n, -1, -1, -1),
            next_obstacle_features=obstacle_features.expand(n, -1, -1, -1),
            next_available_actions=next_available_actions, td_error=0))
        block_features, binary_features = next_block_features, next_binary_features
        if log_images:
            images.append(dict(succ_block_features=next_block_features[0].cpu().numpy()))
    return transitions, images


def track_run_sinks(values, step, context, aim_run=None, wandb_run=None):
    """The two logging sinks of log_episode (successor_dqn.py:544-565 of the reference), shared by the single-env loop
    (step = episode) and the vectorised loop (step = finished episodes so far): aim gets one ``track(value, name=key,
    step=step, context={'context': context})`` per value that is not None; wandb one ``log`` of a dict with the
    reference's keys -- ``episode`` first, then the values as they are (``avg_loss`` may be None),
    ``episode_<step>_combined_image`` (None: this build draws no matplotlib figure) and ``eval_reward`` (the linear
    reward in an evaluation context, else None).  The reference calls the module-level ``wandb.log``; a run object's
    ``log`` is the same call."""
    if aim_run is not None:
        for k, v in values.items():
            if v is not None:
                aim_run.track(v, name=k, step=step, context=dict(context=context))
    if wandb_run is not None:
        payload = {"episode": step}
        payload.update({k: v for k, v in values.items() if k != 'epsilon'})
        payload[f"episode_{str(step).zfill(5)}_combined_image"] = None
        payload["eval_reward"] = values.get('lin_reward') if context == 'evaluation' else None
        wandb_run.log(payload)


def log_episode(episode, transitions, losses, gamma, context='training', policy=None, images=None, log_images=False,
                wandb_run=None, aim_run=None, verbose=False):
    """Episode summary (successor_dqn.py:479-567) without the matplotlib figure; aim / wandb sinks are used when the
    caller passes live run objects."""
    info = {
        'reward': sum(gamma ** i * t.reward for i, t in enumerate(transitions)).item(),
        'lin_reward': sum(gamma ** i * t.lin_reward for i, t in enumerate(transitions)).item(),
        'avg_loss': sum(losses) / len(losses) if losses else None,
        'num_steps': len(transitions),
        'stable': transitions[-1].next_binary_features[0, 0].item(),
        'collision': transitions[-1].next_binary_features[0, 1].item(),
    }
    if policy is not None and hasattr(policy, 'epsilon'):
        info['epsilon'] = policy.epsilon
    track_run_sinks(info, episode, context, aim_run=aim_run, wandb_run=wandb_run)
    return info, None


# ---- CLI (successor_dqn.py:570-787) -----------------------------------------------------------------------------
def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--num_episodes", type=int, default=1000, help="Number of episodes for training.")
    p.add_argument("--max_steps", type=int, default=10, help="Max steps per episode.")
    p.add_argument("--seed", type=int, default=None, help="Random seed.")
    p.add_argument("--num_training_steps", type=int, default=20, help="Number of training steps in each episode.")
    p.add_argument("--learning_rate", type=float, default=0.01, help="Adam learning rate.")
    p.add_argument("--loss_function", choices=['mse_q_values', 'mse_block_features', 'mse_q_values+mse_block_features'],
                   default='mse_q_values', help="Loss function for training")
    p.add_argument("--tau", type=float, default=0.01, help="Step size for updating target net.")
    p.add_argument("--batch_size", type=int, default=32, help="Batch size.")
    p.add_argument("--gamma", type=float, default=0.8, help="Discount factor.")
    p.add_argument("--model", choices=['SuccessorMLP', 'ConvNet', 'UNet'], default='UNet', help="Model type.")
    p.add_argument("--device", choices=['cpu', 'cuda'], default='cuda', help="Device to use (this build is GPU-only).")
    p.add_argument("--image_size", type=parse_img_size, default="64x64", help="Size of image features {width}x{height}.")
    p.add_argument("--load_checkpoint", type=str, default=None, help="Path to a checkpoint to load.")
    p.add_argument("--save_checkpoint", type=str, default=None, help="Path to save the checkpoint.")
    p.add_argument("--checkpoint_every", type=int, default=1000, help="")
    p.add_argument("--evaluate_every", type=int, default=100, help="")
    p.add_argument("--aim", action='store_true', help="Use aim logging.")
    p.add_argument("--aim_repo", type=str, default='aim-data/', help="Path to aim repository.")
    p.add_argument("--bridge_length", type=int, default=1, help="Length of the bridge in blocks")
    p.add_argument("--tower_height", type=int, default=None,
                   help="Height of the tower: bridge_setup(num_stories=N) (README / BASELINE.json flag).")
    p.add_argument("--verbose", action='store_true', help="Verbose output.")
    p.add_argument("--log_images", action='store_true', help="Log images.")
    p.add_argument("--replay_buffer_capacity", type=int, default=2000, help="Replay buffer capacity.")
    p.add_argument("--wandb", type=bool, default=False, help="Use wandb logging.")
    p.add_argument("--num_envs", type=int, default=1, help="Environments advanced in lock-step on the GPU.")
    p.add_argument("--prioritized_replay", action='store_true',
                   help="Sample transitions in proportion to |td_error| + 1e-5 (PrioritizedReplayBuffer, replay_memory.py:45-93).")
    p.add_argument("--shapes", choices=['trapezoid', 'hexagon', 'both'], default='trapezoid')
    return p


def make_setup_fct(args):
    trap, hexa = args['shapes'] in ('trapezoid', 'both'), args['shapes'] in ('hexagon', 'both')
    if args.get('tower_height'):
        return lambda: bridge_setup(num_stories=args['tower_height'], trapezoid=trap, hexagon=hexa)
    return lambda: horizontal_bridge_setup(num_obstacles=args['bridge_length'], trapezoid=trap, hexagon=hexa)


def make_nets(args, device):
    if args['model'] == 'SuccessorMLP':
        mk = lambda: SuccessorMLP(img_size=args['image_size'], hidden_dims=[256, 128, 64, 128, 256])
    elif args['model'] == 'ConvNet':
        mk = lambda: ConvNet(img_size=args['image_size'])
    elif args['model'] == 'UNet':
        mk = Policy
    else:
        raise ValueError(f"Unknown model type {args['model']}.")
    policy_net, target_net = mk().to(device), mk().to(device)
    policy_net.apply(init_weights)
    target_net.load_state_dict(policy_net.state_dict())
    flatten_nets(policy_net, target_net)
    return policy_net, target_net


def main(argv=None):
    args = vars(build_parser().parse_args(argv))
    if args['device'] == 'cpu':
        raise SystemExit("this build has no CPU path: the simulator and the DQN ops are HIP kernels (use --device cuda)")
    from bridges_hip import abi
    abi.require_gpu()
    local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    if args['seed'] is not None:
        random.seed(args['seed'])
        np.random.seed(args['seed'])
        torch.manual_seed(args['seed'])
    aim_run = wandb_run = None
    if args['aim']:                                                         # successor_dqn.py:671-674
        try:
            import aim
        except ImportError as e:
            raise SystemExit("--aim was given but the 'aim' package is not installed") from e
        aim_run = aim.Run(experiment="SuccessorQLearning", repo=args['aim_repo'])
    if args['wandb']:
        try:
            import wandb
        except ImportError as e:
            raise SystemExit("--wandb was given but the 'wandb' package is not installed") from e
        wandb_run = wandb.init(project="dual_arm", config=args)
    if args['num_envs'] > 1:
        from robotoddler.training.vec_dqn import run_vectorised
        return run_vectorised(args, device, aim_run=aim_run, wandb_run=wandb_run)

    x_discr_ground = np.linspace(-2, 0, 10)
    offset_values = [0]
    xlim, ylim = (-3, 7), (0., 10)
    gamma = args['gamma']
    policy_net, target_net = make_nets(args, device)
    if args['prioritized_replay']:
        replay_buffer = PrioritizedReplayBuffer(capacity=args['replay_buffer_capacity'], gamma=gamma, policy_net=policy_net,
                                                target_net=target_net, device=device)
    else:
        replay_buffer = ReplayBuffer(capacity=args['replay_buffer_capacity'])
    eps_greedy = EpsilonGreedy(eps_start=0.5, gamma=0.999, eps_end=0.05, episode=0, max_steps=args['max_steps'], device=device)
    greedy = lambda q, *a, **k: torch.argmax(q)
    setup_fct = make_setup_fct(args)
    env = AssemblyGym(reward_fct=sparse_reward, max_steps=args['max_steps'], restrict_2d=True,
                      assembly_env=AssemblyEnv(render=False))
    optimizer = torch.optim.Adam(policy_net.parameters(), lr=args['learning_rate'])
    first_episode = 1
    if args['load_checkpoint']:                          # successor_dqn.py:654-665 (disabled there: "not tested")
        from robotoddler.utils.utils import load_checkpoint, optimizer_to
        meta = load_checkpoint(args['load_checkpoint'], policy_net, target_net, replay_buffer, optimizer,
                               devices=dict(policy_net=device, target_net=device, optimizer=device))
        optimizer_to(optimizer, device)
        first_episode = int(meta['episode']) + 1
        eps_greedy = EpsilonGreedy(eps_start=0.5, gamma=0.999, eps_end=0.05, episode=int(meta['episode']),
                                   max_steps=args['max_steps'], device=device)
    history = []
    roll = dict(env=env, policy_net=policy_net, setup_fct=setup_fct, x_discr_ground=x_discr_ground, xlim=xlim, ylim=ylim,
                offset_values=offset_values, img_size=args['image_size'], device=device, log_images=False)
    for i in range(first_episode, args['num_episodes'] + 1):
        transitions, images = rollout_episode(policy=eps_greedy.step(), **roll)
        replay_buffer.push(transitions)
        losses = train_policy_net(policy_net=policy_net, target_net=target_net, optimizer=optimizer,
                                  loss_fct=args['loss_function'], replay_buffer=replay_buffer, gamma=gamma,
                                  batch_size=args['batch_size'], n_steps=args['num_training_steps'], device=device)
        update_target_net(policy_net=policy_net, target_net=target_net, tau=args['tau'])
        log_info, _ = log_episode(episode=i, transitions=transitions, policy=eps_greedy, losses=losses, gamma=gamma,
                                  aim_run=aim_run, wandb_run=wandb_run)
        history.append(log_info)
        if args['verbose']:
            print(f"episode {i}: {log_info}")
        if args['save_checkpoint'] and i % args['checkpoint_every'] == 0:     # utils.py:54-89 layout
            sync_fused_optimizer(policy_net)
            save_checkpoint(args['save_checkpoint'], policy_net, target_net, replay_buffer, optimizer, i,
                            {k: (str(v) if not isinstance(v, (int, float, str, bool, type(None))) else v) for k, v in args.items()})
        if i % args['evaluate_every'] == 0:
            transitions, _ = rollout_episode(policy=greedy, **roll)
            ev, _ = log_episode(episode=i, transitions=transitions, losses=None, context='evaluation', gamma=gamma,
                                aim_run=aim_run, wandb_run=wandb_run)
            if args['verbose']:
                print(f"evaluation {i}: {ev}")
    return history


if __name__ == '__main__':
    main()
