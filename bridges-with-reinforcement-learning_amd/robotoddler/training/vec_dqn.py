"""Vectorised successor-feature DQN: E environments per GPU in lock-step (VecAssemblyGym), device replay ring of
compact records re-rasterised on sample, fused HIP target / soft-update ops, one all-gather of records per
lock-step across ranks.

Semantics follow rollout_episode / train_policy_net / update_target_net of the reference
(robotoddler/training/successor_dqn.py:157-288, 365-475) with these batched readings:
* epsilon-greedy draws one uniform per env; exploring envs take the candidate whose raster overlaps least with the
  per-episode-step count images (successor_dqn.py:112-132), all envs scored against the images of the lock-step
  start and the chosen rasters added afterwards;
* the TD target is the elementwise  lin_reward + gamma * q'  (the single-env reference path trains on a [B,B]
  broadcast of it because its lin_reward is [B,1]; that quirk is reproduced only in the single-env loop);
* "done" of a transition = terminated | truncated | no next action (successor_dqn.py:393, 409-411);
* epsilon decays once per LOCK-STEP (the reference: once per episode, successor_dqn.py:702 -- with thousands of envs
  hundreds of episodes end per lock-step and a per-episode decay would reach the floor within a dozen lock-steps);
* uniform replay draws WITH replacement (torch.randint on the device ring; the reference's random.sample draws without:
  a duplicate inside a batch of 32 out of >= 10^4 records has probability ~5e-2 and only repeats a sample);
* with several ranks every rank trains an identical replica on the identical gathered ring (same sampling seed); extra
  GPUs add rollout throughput, not optimiser throughput; parameters are re-broadcast every 100 lock-steps.

How a lock-step is spent: acting through the factored SuccessorMLP forward whose first layer reads
the bit-packed rasters (bridges_bits_linear); the TD / successor-feature targets of ALL optimiser steps of the lock-step
in one pass (the target net is constant meanwhile); the optimiser steps themselves as replays of one HIP graph.
"""
import os
import time
import warnings

import numpy as np
import torch

from bridges_hip import dqn_ops, ops
from bridges_hip.shapes import load_urdf
from bridges_hip.vec_env import VecAssemblyGym
from robotoddler.training import distributed as D
from robotoddler.training import records as R


class VecDQN:
    def __init__(self, policy_net, target_net, optimizer, env, replay_capacity, batch_size, gamma, tau, loss_function,
                 seed=0, rank=0, eps_start=0.5, eps_end=0.05, eps_decay=0.999, prioritized=False):
        self.policy_net, self.target_net, self.opt, self.env = policy_net, target_net, optimizer, env
        self.device = env.device
        self.B, self.gamma, self.tau = batch_size, gamma, tau
        self.loss_parts = loss_function.split('+')
        self.prioritized = bool(prioritized)      # PrioritizedReplayBuffer semantics (replay_memory.py:45-93)
        self.ring = R.ReplayRing(replay_capacity, self.device)
        # replay sampling must be identical on every rank (replicated rings) -> shared seed; exploration differs
        self.sample_gen = torch.Generator(device=self.device).manual_seed(1234567 + seed)
        self.explore_gen = torch.Generator(device=self.device).manual_seed(7654321 + seed * 1000 + rank)
        self.seed, self.rank = int(seed), int(rank)
        self.epsilon, self.eps_end, self.eps_decay = eps_start, eps_end, eps_decay
        self.step_images = torch.zeros((env.K + 1, env.img, env.img), dtype=torch.float32, device=self.device)
        # scratch env used to rebuild the candidate sets of sampled next states (see _replay_env)
        self.replay_env = VecAssemblyGym(batch_size, env.shapes, env.obstacles, env.targets, max_steps=env.max_steps,
                                         mu=env.mu, density=env.density, bounds=env.bounds, xlim=env.xlim,
                                         ylim=env.ylim, x_discr_ground=env.x_discr_ground,
                                         offset_values=env.offset_values, device=self.device, a_max=env.a_max,
                                         img_size=(env.img, env.img), f32_rasters=self._replay_f32())
        self.mse = torch.nn.MSELoss()
        for g in optimizer.param_groups:                # step counter on the device: the train step is graph-captured
            if 'capturable' in g:
                g['capturable'] = True
        self._graph_state, self._eager_calls = None, 0
        self._eager_reduce = None                            # dqn_ops.ReduceTables of the eager optimiser steps
        self.episodes_done = 0
        self.env_steps = 0
        self._counts_host = torch.zeros(2, dtype=torch.int64).pin_memory()      # (env-steps, finished episodes) of a lock-step

    ROW_CHUNK = 2048       # rows per forward call: ONE input shape for the whole run (MIOpen tunes per shape)
    DEDUP_STATES = True    # envs in the same state share one set of candidate rows (tests compare with False)
    DEDUP_ROWS = True      # conv nets: feed every distinct (state, candidate, stable flag) input once (tests compare with False)
    TRACK_ROWS = False     # tools: count the valid rows of every Q pass (rows_seen) beside the rows actually fed (rows_fed)

    def _rows(self, env, stable):
        """The candidate rows a Q pass over ``env`` is fed: (idx, row_env, (seg_lo, seg_hi), rep).  Thousands of envs of one task
        pass through the same states -- every freshly reset env holds the empty assembly, and the reference's policy is a
        near-deterministic function of the state (greedy arg-max, or the least-tried candidate of the episode step) -- and
        the rows of a state and their values depend on the state alone: envs whose block lists and stable flag are equal
        word for word (VecAssemblyGym.state_groups) share the rows of the first of them; rows seg_lo[e] .. seg_hi[e] are env
        e's.  Exact (the same kernels on the same inputs), two extra launches, no extra wait."""
        cache = getattr(env, "_dqn_rows", None)
        if cache is not None and cache[0] == env._cand_version:
            return cache[1]
        rep = env.state_groups(stable) if self.DEDUP_STATES else None
        idx, row_env = env.valid_rows(rep)
        out = (idx, row_env, env.valid_segments(), rep)
        env._dqn_rows = (env._cand_version, out)
        if self.TRACK_ROWS:
            seen = env.n_valid[:env.E].sum()
            self._rows_seen_dev = seen if getattr(self, "_rows_seen_dev", None) is None else self._rows_seen_dev + seen
        return out

    @property
    def rows_seen(self):
        dev = getattr(self, "_rows_seen_dev", None)
        return int(dev) if dev is not None else 0

    def _distinct_rows(self, env, idx, row_env, stable_flag):
        """Candidate rows whose network input is the same tensor, bit for bit: the input of a row is (state raster of its env,
        its candidate raster, the env's stable flag, the task's reward map and obstacle raster), and thousands of environments
        of one task pass through the same early states -- every freshly reset env holds the same state and the same
        candidates.  Rows are keyed by a 64-bit hash of their bit rasters, grouped with torch.unique, and every row is then
        compared WORD FOR WORD with the representative of its group (a hash collision -- probability ~ n^2 / 2^64 -- sends the
        call down the plain path), so feeding only the representatives and copying their outputs is exact.
        -> (rep [m]: positions into idx of one row per distinct input, inverse [n]: group of every row) or None."""
        n = idx.numel()
        if not self.DEDUP_ROWS or n < 2:
            return None
        m = getattr(self, "_hash_mult", None)
        if m is None:
            g = torch.Generator().manual_seed(0x5eed5)
            m = self._hash_mult = (torch.randint(-2 ** 62, 2 ** 62, (2, 64), generator=g, dtype=torch.int64) | 1).to(self.device)
        cb = env.cand_bits.index_select(0, idx)                                    # [n, 64] int64
        flag = stable_flag.to(torch.int64)
        hs = (env.state_bits * m[0]).sum(dim=1) + flag * 0x51ED270B27B4F3D           # [E]  (int64 arithmetic wraps)
        key = hs.index_select(0, row_env) * 0x2545F4914F6CDD1D + (cb * m[1]).sum(dim=1)
        key = key ^ (key >> 29)
        uniq, inverse = torch.unique(key, return_inverse=True)                      # (one wait: the number of groups)
        n_groups = uniq.numel()
        if n_groups == n:
            return None
        pos = torch.arange(n, device=self.device)
        rep = torch.full((n_groups,), n, dtype=torch.int64, device=self.device).scatter_reduce_(0, inverse, pos, reduce="amin")
        r = rep.index_select(0, inverse)                                             # representative of every row
        renv = row_env.index_select(0, r)
        same = ((cb == cb.index_select(0, r)).all(dim=1)
                & (env.state_bits.index_select(0, row_env) == env.state_bits.index_select(0, renv)).all(dim=1)
                & (flag.index_select(0, row_env) == flag.index_select(0, renv)))
        if not bool(same.all()):
            return None
        return rep, inverse

    def _forward_rows(self, net, env, idx, row_env, stable_flag):
        """net(...) over the candidate rows in chunks of ROW_CHUNK rows; the last chunk is padded with copies of row 0
        (sliced off again), so the convolution / GEMM shapes never change from lock-step to lock-step.  Rows with identical
        inputs are fed once (_distinct_rows).  -> (q [n], successor block features of the DISTINCT rows or None, successor
        binary features of the distinct rows or None, inverse [n] = the distinct row of every row (None: every row is fed))."""
        groups = self._distinct_rows(env, idx, row_env, stable_flag)
        inverse = None
        if groups is not None:
            rep, inverse = groups
            idx, row_env = idx.index_select(0, rep), row_env.index_select(0, rep)
        n, C = idx.numel(), self.ROW_CHUNK
        self.rows_fed = getattr(self, "rows_fed", 0) + n
        pad = (-n) % C
        if pad:
            idx = torch.cat([idx, idx[:1].expand(pad)])
            row_env = torch.cat([row_env, row_env[:1].expand(pad)])
        outs = [net(*self._row_features(env, idx[o:o + C], row_env[o:o + C], stable_flag)) for o in range(0, n + pad, C)]
        q = torch.cat([o[0] for o in outs])[:n]
        sf = torch.cat([o[1] for o in outs])[:n] if outs[0][1] is not None else None
        sb = torch.cat([o[2] for o in outs])[:n] if outs[0][2] is not None else None
        if inverse is not None:
            q = q.index_select(0, inverse)
        return q, sf, sb, inverse

    # ------------------------------------------------------------------ features of the rows a net is fed
    def _row_features(self, env, idx, row_env, stable_flag):
        n = idx.numel()
        if env.cand_raster is not None:
            block = env.crop(env.state_raster[row_env]).unsqueeze(1)     # crop: no-op for the 64x64 default
            action = env.crop(env.cand_raster[idx]).unsqueeze(1)
        else:                                                            # env without f32 rasters: expand the rows asked for
            block = env.crop(ops.bits_to_f32(env.state_bits[row_env])).unsqueeze(1)
            action = env.crop(ops.bits_to_f32(env.cand_bits[idx])).unsqueeze(1)
        binary = torch.zeros((n, 6), dtype=torch.float32, device=self.device)
        binary[:, 0] = stable_flag[row_env].float()
        reward = env.reward_features.unsqueeze(0).expand(n, -1, -1, -1)
        obstacle = env.obstacle_raster.unsqueeze(0).expand(n, -1, -1, -1)
        return block, binary, action, reward, obstacle

    FACTORED_ACTING = True      # tests set it to False to run the same lock-steps through the plain module forward

    @classmethod
    def _factored(cls, net):
        """Acting through the factored SuccessorMLP forward on bit-packed rasters.  The bit-packed first layer is built for
        64x64 images; other --image_size values (and every other net) act through the module forward on f32 rasters."""
        return (cls.FACTORED_ACTING and hasattr(net, "q_from_first_layer")
                and tuple(getattr(net, "img_size", (64, 64))) == (64, 64))

    @classmethod
    def acting_needs_f32_rasters(cls, net):
        """False when acting reads only the bit-packed rasters of the rollout env: it can then be created with
        f32_rasters=False and its rasteriser skips the 16 KiB-per-candidate expansion."""
        return not cls._factored(net)


    @staticmethod
    def _stable_flags(env):
        """'stable' binary feature of the current state of every env (a freshly reset env is stable)."""
        fresh = env.n_blocks == 0
        return torch.where(fresh, torch.ones_like(fresh), env.step_flags[:, 1].bool())

    @torch.no_grad()
    def _policy_q(self, env, idx, row_env, stable):
        """q of the policy net for the candidate rows ``idx`` of ``env``."""
        return self._net_q(self.policy_net, env, idx, row_env, stable)

    @torch.no_grad()
    def _net_q(self, net, env, idx, row_env, stable, return_h=False):
        """q of ``net`` for the candidate rows; with return_h (factored nets only) also the first layer's pre-activation
        of every row, from which the successor features of selected rows follow without a second first-layer pass."""
        net.eval()
        if not self._factored(net):
            if env.cand_raster is None:
                raise ValueError("this Q-network acts on f32 rasters: create the rollout env with f32_rasters=True")
            return self._forward_rows(net, env, idx, row_env, stable)[0]
        # the first layer consumes the BIT-PACKED rasters (bridges_bits_linear): a raster times a weight slice is the
        # sum of the ~35 weight rows of its set pixels, so neither f32 images nor a [n, 4096] GEMM
        px = 64 * 64
        self.rows_fed = getattr(self, "rows_fed", 0) + idx.numel()
        W1 = net.first_layer().weight
        # the part of the first layer that does not depend on the images: a two-row table indexed by the env's stable flag
        # (the other binary features are 0, as in the reference without pybullet)
        ro = getattr(env, "_reward_obstacle_flat", None)
        if ro is None:
            ro = env._reward_obstacle_flat = torch.cat([env.reward_features.reshape(-1), env.obstacle_raster.reshape(-1)]).contiguous()
        base = ops.bits_linear(env.state_bits, W1[:, :px].T, base=net.first_layer_stable_table(ro), base_row=stable.long())
        h_pre = ops.bits_linear(env.cand_bits, W1[:, px:2 * px].T, bits_row=idx, base=base, base_row=row_env)
        q = net.q_from_first_layer(h_pre, env.reward_features, head=ops.sigmoid_dot, fused_head=ops.head_sigmoid_dot)
        return (q, h_pre) if return_h else q

    @torch.no_grad()
    def td_errors(self, rec):
        """td_error of the transitions just recorded (successor_dqn.py:413-426): |q(s,a) - (reward + 0.95 max_a' q(s',a'))|
        with the policy net as it is now, next value 0 when done; the env holds s' (or a fresh state when done)."""
        env, E = self.env, self.env.E
        done = rec[:, R.O_DONE] > 0.5
        stable = self._stable_flags(env)
        idx, row_env, seg, _rep = self._rows(env, stable)
        next_q = torch.zeros(E, dtype=torch.float32, device=self.device)
        if idx.numel():
            q = self._policy_q(env, idx, row_env, stable)
            zeros = torch.zeros(E, dtype=torch.float32, device=self.device)
            next_q, _, _ = dqn_ops.td_target(seg, q.contiguous().float(), zeros, done | (env.n_valid[:E] == 0), 1.0)   # segmented max
        expected = rec[:, R.O_REWARD].float() + 0.95 * next_q                      # hard-coded 0.95 (successor_dqn.py:425)
        return (self._q_sel - expected).abs()

    # ------------------------------------------------------------------ one lock-step of acting
    @torch.no_grad()
    def act(self, greedy=False):
        env, E = self.env, self.env.E
        stable = self._stable_flags(env)
        idx, row_env, seg, rep = self._rows(env, stable)
        self._q_sel = torch.zeros(E, dtype=torch.float32, device=self.device)
        if idx.numel():
            step_of_row = env.n_blocks[row_env].long()
            q = self._policy_q(env, idx, row_env, stable)
            if self._factored(self.policy_net):
                # overlap of every candidate with the count image of its episode step, straight from the bit-packed
                # rasters (exact: integer-valued sums)
                join = ops.bits_dot(env.cand_bits, self.step_images, step_of_row, bits_row=idx)
            else:
                join = (self.step_images[step_of_row] * env.crop(env.cand_raster[idx])).sum(dim=(1, 2))
            # greedy row = first maximum of q, exploring row = first minimum of the overlap, per env, in ONE launch
            # (bridges_eps_greedy_select; an `if explore.any()` here would make the host wait for the Q pass it has just queued)
            u = torch.rand(E, generator=self.explore_gen, device=self.device)
            sel_compact, sel_index, self._q_sel, ex_w = ops.eps_greedy_select(seg, q, join, u, self.epsilon, greedy, idx,
                                                                              env.cand_offset[:E], rep=rep)
            # count images of the explored choices; every env takes part with weight 0 or 1, so no host decision
            step_of_env = env.n_blocks.long()
            if env.img == 64:                                     # the set pixels of the chosen rasters, by float atomics
                ops.bits_accumulate_(self.step_images, env.cand_bits, step_of_env, weight=ex_w, bits_row=sel_compact)
            else:
                picked = env.crop(ops.bits_to_f32(env.cand_bits[sel_compact])) * ex_w[:, None, None]
                self.step_images.index_add_(0, step_of_env, picked)
        else:
            sel_compact = torch.zeros(E, dtype=torch.long, device=self.device)
            sel_index = (sel_compact - env.cand_offset[:E].long()).clamp(min=0).to(torch.int32)
        # the record of the lock-step in two launches around the step (bridges_record_state / _result); the torch formulation
        # R.snapshot + R.make_records (~35 launches) is what tests/test_gpu_vec_dqn.py compares them with
        rec = R.pack_state(env, sel_compact)
        env.step(sel_index)
        valid = R.pack_result(env, rec)
        return rec, valid

    # ------------------------------------------------------------------ gradient steps on sampled batches
    def _replay_env(self, n_states):
        """Scratch env that rebuilds the states / candidate sets of sampled transitions (grown on demand)."""
        if self.replay_env.E < n_states:
            env = self.env
            self.replay_env = VecAssemblyGym(n_states, env.shapes, env.obstacles, env.targets, max_steps=env.max_steps,
                                             mu=env.mu, density=env.density, bounds=env.bounds, xlim=env.xlim,
                                             ylim=env.ylim, x_discr_ground=env.x_discr_ground,
                                             offset_values=env.offset_values, device=self.device, a_max=env.a_max,
                                             img_size=(env.img, env.img), f32_rasters=self._replay_f32())
        return self.replay_env

    def _replay_f32(self):
        """The scratch env writes f32 rasters of every raw candidate only for nets that consume them row by row; the
        factored MLP reads the bit rasters and expands the few rows it needs (the arg-max row of each transition)."""
        return not self._factored(self.target_net)

    @torch.no_grad()
    def _targets(self, rec):
        """Inputs and TD targets of the transitions in ``rec`` (train_policy_net, successor_dqn.py:178-213).  The
        target net is constant during one train_policy_net call (it is only updated afterwards, :704-708), so the
        targets of all its n_steps batches can be computed in one pass: one state rebuild, one candidate refresh,
        one target-net forward and one k_td_target launch for n_steps * batch_size transitions."""
        n = rec.shape[0]
        renv = self._replay_env(n)
        E = renv.E
        # state s' (= s + action block): candidates, masks, rasters by the same kernels as the rollout; s is the
        # prefix of its block list, so its raster comes out of the same per-block bit rasters.  One launch unpacks the
        # records into the scratch env (envs beyond n repeat record 0 and are sliced off below); R.unpack_states +
        # load_states + prefix_state_bits is the torch formulation the tests compare it with.
        bits_s, lin, stable_s, done_rec, stable_n = renv.load_records(rec.contiguous())
        block_f = renv.crop(ops.bits_to_f32(bits_s)).unsqueeze(1)
        action_f = renv.crop(ops.bits_to_f32(renv.state_bits & ~bits_s)).unsqueeze(1)         # s' minus s = the new block
        stable_n = stable_n.bool()
        idx, row_env, seg, _rep = self._rows(renv, stable_n)      # transitions with the same next state share its rows
        done = done_rec.bool() | (renv.n_valid[:E] == 0)
        use_sf = 'mse_block_features' in self.loss_parts
        if idx.numel() and self._factored(self.target_net):
            # q of every next candidate through the factored forward on the bit-packed rasters; the 8204-wide output
            # (successor features) is only needed for the arg-max row of each transition
            nq, h_pre = self._net_q(self.target_net, renv, idx, row_env, stable_n, return_h=True)
            nq = nq.contiguous().float()
            q_target, _, arg = dqn_ops.td_target(seg, nq, lin, done, self.gamma)
            sf_target = None
            if use_sf:
                best = arg.long().clamp_(0, idx.numel() - 1)                       # empty segments are 'done': row unused
                # channel 0 of the arg-max rows' successor features from the first-layer pre-activations already at hand
                nsf0 = self.target_net.sf0_from_first_layer(h_pre.index_select(0, best)).contiguous()
                one_each = torch.arange(E + 1, dtype=torch.int32, device=self.device)
                _, sf_target, _ = dqn_ops.td_target(one_each, nq[best].contiguous(), lin, done, self.gamma,
                                                    next_sf=nsf0, action_raster=action_f.squeeze(1))
        elif idx.numel():
            nq, nsf, _, inverse = self._forward_rows(self.target_net, renv, idx, row_env, stable_n)
            if use_sf and nsf is None:
                raise ValueError("No successor block features available from the chosen policy net.")
            nq = nq.contiguous().float()
            q_target, _, arg = dqn_ops.td_target(seg, nq, lin, done, self.gamma)
            sf_target = None
            if use_sf:
                # successor features of the arg-max row of every transition only (nsf holds the DISTINCT rows' outputs)
                best = arg.long().clamp_(0, idx.numel() - 1)                       # empty segments are 'done': row unused
                nsf0 = nsf[:, 0].index_select(0, best if inverse is None else inverse.index_select(0, best)).reshape(E, -1).contiguous()
                one_each = torch.arange(E + 1, dtype=torch.int32, device=self.device)
                _, sf_target, _ = dqn_ops.td_target(one_each, nq[best].contiguous(), lin, done, self.gamma,
                                                    next_sf=nsf0, action_raster=action_f.squeeze(1))
        else:
            q_target = lin
            sf_target = action_f.reshape(E, -1) if use_sf else None
        binary = torch.zeros((E, 6), dtype=torch.float32, device=self.device)
        binary[:, 0] = stable_s
        return block_f[:n], binary[:n], action_f[:n], q_target[:n], (sf_target[:n] if use_sf else None)

    def _loss(self, q, sf, q_target, sf_target):
        loss = 0.
        if 'mse_q_values' in self.loss_parts:
            loss = loss + self.mse(q, q_target)
        if sf_target is not None:
            loss = loss + self.mse(sf[:, 0], sf_target.view_as(sf[:, 0]))
        return loss

    def _capture_train_graph(self, n_max, use_sf):
        """One optimiser step (batch gather, forward, loss, backward, Adam) as a HIP graph.  The batch is gathered
        inside the graph from the static arrays of all batches of the lock-step by a device-side step counter, so a
        train step is exactly one graph launch (eager PyTorch needs ~60 launches of a few microseconds of work each
        and is bound by their launch latency)."""
        B, dev = self.B, self.device
        px = (self.env.img, self.env.img)
        st = dict(block=torch.zeros((n_max * B, 1, *px), device=dev), binary=torch.zeros((n_max * B, 6), device=dev),
                  action=torch.zeros((n_max * B, 1, *px), device=dev), q=torch.zeros(n_max * B, device=dev),
                  sf=torch.zeros((n_max * B, px[0] * px[1]), device=dev) if use_sf else None,
                  counter=torch.zeros((), dtype=torch.int64, device=dev),
                  losses=torch.zeros(n_max, device=dev),
                  lane=torch.arange(B, device=dev), iota=torch.arange(n_max, device=dev), n_max=n_max, use_sf=use_sf)
        reward = self.env.reward_features.unsqueeze(0).expand(B, -1, -1, -1)
        obstacle = self.env.obstacle_raster.unsqueeze(0).expand(B, -1, -1, -1)
        st["fused"] = self._fused_step_enabled()

        def body_fused():
            # forward, losses and backward of the MLP as ~20 hand-written launches on the f32 matrix cores
            # (bridges_hip/mlp_ops.py, csrc/mlp_kernels.hip) instead of ~60 library / element-wise ones; the gradients land
            # in the parameters' .grad, the optimiser is torch's fused Adam as before.  Same losses as _loss.
            n = n_max * B
            st["step"].launch(st["counter"], st["block"].view(n, -1), st["action"].view(n, -1), st["binary"], st["reward"],
                              st["obstacle"], st["q"], st["sf"], st["losses"])
            if not st["step"].fused_adam:                 # else the Adam update is the last launch of the sequence
                self.opt.step()

        def body():
            idx = st["lane"] + st["counter"] * B
            q, sf, _ = self.policy_net(st["block"].index_select(0, idx), st["binary"].index_select(0, idx),
                                       st["action"].index_select(0, idx), reward, obstacle)
            # same losses as _loss, but the 131 072-element mean is reduced row-wise and then over the 32 rows: the
            # multi-workgroup (semaphore) reduction nn.MSELoss launches for it returned garbage on some replays
            # (negative "MSE", ROCm 7.2 + torch 2.10; eager never) while every single-workgroup reduction was right.
            # The value is logged through a one-hot of the step counter (pure elementwise arithmetic).
            loss = 0.
            if 'mse_q_values' in self.loss_parts:
                loss = loss + ((q - st["q"].index_select(0, idx)) ** 2).mean()
            if use_sf:
                loss = loss + ((sf[:, 0].reshape(B, -1) - st["sf"].index_select(0, idx)) ** 2).mean(dim=1).mean()
            st["losses"].add_((st["iota"] == st["counter"]).to(torch.float32) * loss.detach())
            with dqn_ops.deferred_wgrad_reduce(st["reduce"]):          # the conv layers' weight-gradient reductions as one launch
                loss.backward()
            if st["adam"] is not None:
                st["adam"].step()                          # one launch of 1024-element chunks over all parameter tensors
            else:
                self.opt.step()
            st["counter"].add_(1)

        self.policy_net.train()
        self.opt.zero_grad(set_to_none=True)
        st["adam"] = None
        st["reduce"] = dqn_ops.ReduceTables(dev)
        if not st["fused"]:
            from bridges_hip.dqn_ops import MultiTensorAdam
            try:
                st["adam"] = MultiTensorAdam(self.opt)
            except ValueError:
                pass                                       # another optimiser, or one with options the launch does not cover
        if st["fused"]:
            from bridges_hip.mlp_ops import FusedSuccessorStep
            st["step"] = FusedSuccessorStep(self.policy_net, B, 'mse_q_values' in self.loss_parts, use_sf,
                                            optimizer=self.opt)
            st["reward"] = self.env.reward_features.reshape(-1).contiguous()
            st["obstacle"] = self.env.obstacle_raster.reshape(-1).contiguous()
            # the first layer's input rows of all batches of a call are built by ONE launch before the replays
            # (train_steps), a replayed step reads batch `counter` of them: one launch per optimiser step less
            st["step"].allocate_inputs(n_max)
            st["step"]._prepared = True                  # captured in the form that reads the pre-built rows
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            body_fused() if st["fused"] else body()
        st["graph"] = graph
        # all n_max steps of a call as ONE graph as well (the hand-written step only: its launches read the batch counter from
        # the device, so n_max copies of the sequence are the n_max steps): between two graph launches the GPU idles 8.7 us
        # (rocprofv3 trace of the loop), inside a graph consecutive kernels follow each other without a gap
        st["graph_all"] = None
        if st["fused"] and n_max > 1:
            graph_all = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_all):
                for _ in range(n_max):
                    body_fused()
            st["graph_all"] = graph_all
        return st

    def _fused_step_enabled(self):
        """The hand-written training step applies to SuccessorMLP with the two MSE losses of the CLI
        (BRIDGES_FUSED_MLP_STEP=0: the autograd step inside the graph, as before)."""
        from robotoddler.models.cv import SuccessorMLP
        return (isinstance(self.policy_net, SuccessorMLP) and os.environ.get("BRIDGES_FUSED_MLP_STEP", "1") != "0"
                and set(self.loss_parts) <= {'mse_q_values', 'mse_block_features'}
                and all(p.dtype == torch.float32 for p in self.policy_net.parameters()))

    def _train_graph(self, n_steps, use_sf):
        """The captured train step, or None (eager).  Default: on for SuccessorMLP, whose step has no multi-workgroup
        reduction left in it (see _capture_train_graph) and is verified against the eager run at the bench size
        (tests/test_gpu_vec_dqn.py), and on for ConvNet: its ConvBlocks run forward and backward on the hand-written
        kernels (deterministic partial-sum reductions, csrc/conv_train_kernels.hip), what is left of the library in its step
        are single-workgroup reductions over the 32 batch rows, the eager step is bound by the host (1.2 ms of Python and
        launches for 0.57 ms of GPU work), and the replays run under the on-device restore guard (_guard_restore).  On for
        the U-Net policy as well: its 3x3 convolutions are the hand-written ones, its transposed and 1x1 convolutions stay with
        the library but without their bias, whose gradient -- a sum over 131 k elements that torch reduces with the
        multi-workgroup kernel that once misbehaved inside a replay -- comes from bridges_bias_grad (dqn_ops.conv_bias_train).
        BRIDGES_TRAIN_GRAPH=0/1 overrides.  The first calls always run eagerly (they initialise the optimiser state and the library workspaces a
        capture needs)."""
        from robotoddler.models.cv import ConvNet, Policy, SuccessorMLP
        default = "1" if isinstance(self.policy_net, (SuccessorMLP, ConvNet, Policy)) else "0"
        if os.environ.get("BRIDGES_TRAIN_GRAPH", default) != "1":
            return None
        st = self._graph_state
        if st is not None and (st["n_max"] < n_steps or st["use_sf"] != use_sf):
            self.sync_optimizer_state()
            st = self._graph_state = None                     # more batches per call than captured for: capture again
        if st is None:
            if self._eager_calls < 2:
                self._eager_calls += 1
                return None
            try:
                st = self._graph_state = self._capture_train_graph(n_steps, use_sf)
            except RuntimeError as e:                         # same arithmetic either way: keep training eagerly
                warnings.warn(f"train-step graph capture failed, staying eager: {e}")
                self._eager_calls = -(1 << 30)
                return None
        return st

    def train_steps(self, n_steps, defer=False):
        """n_steps optimiser steps on n_steps independently sampled batches; returns the losses (one host sync).
        defer=True: returns a ``DeferredLosses`` -- the losses travel to pinned host memory behind the optimiser steps and
        ``.get()`` waits for that copy only, so the host can queue the next lock-step's acting while the GPU still
        trains (the list form makes the host wait for the last optimiser step before it queues anything)."""
        if len(self.ring) < self.B or n_steps <= 0:
            return DeferredLosses(None, None, self) if defer else []
        B = self.B
        # n_steps independent batches = ONE draw of n_steps * B records: both sampling rules draw with replacement, so
        # the batches are i.i.d. either way (25 separate draws cost ~100 launches of host time per lock-step)
        rec = self.ring.sample(n_steps * B, self.sample_gen, self.prioritized)
        block_f, binary, action_f, q_target, sf_target = self._targets(rec)
        use_sf = sf_target is not None
        st = self._train_graph(n_steps, use_sf)
        if st is not None:
            n = n_steps * B
            if st.get("fused"):
                st["step"].check_hyperparameters()
                # the first layer's input rows of all n_steps batches in one launch, straight from the target pass's tensors
                # (no staging copy of the 13 MB block / action images: a replayed step reads only x_all, q and sf)
                st["step"].prepare_inputs(n_steps, block_f.reshape(n, -1).contiguous(), action_f.reshape(n, -1).contiguous(),
                                          binary.contiguous(), st["reward"], st["obstacle"])
            else:
                if st["adam"] is not None:
                    st["adam"].check_hyperparameters()
                st["block"][:n].copy_(block_f); st["binary"][:n].copy_(binary); st["action"][:n].copy_(action_f)
                self._guard_snapshot(st)
            st["q"][:n].copy_(q_target)
            if use_sf:
                st["sf"][:n].copy_(sf_target.reshape(n, -1))
            st["counter"].zero_()
            st["losses"].zero_()
            if n_steps == st["n_max"] and st["graph_all"] is not None:
                st["graph_all"].replay()
            else:
                for _ in range(n_steps):
                    st["graph"].replay()
            if not st.get("fused"):
                self._guard_restore(st, st["losses"][:n_steps])
            if defer:
                host = torch.empty(n_steps, dtype=torch.float32, pin_memory=True)
                host.copy_(st["losses"][:n_steps], non_blocking=True)
                done = torch.cuda.Event()
                done.record()
                return DeferredLosses(host, done, self)
            return self._check_graph_losses(st["losses"][:n_steps].tolist())            # the one host sync of the call
        reward = self.env.reward_features.unsqueeze(0).expand(B, -1, -1, -1)
        obstacle = self.env.obstacle_raster.unsqueeze(0).expand(B, -1, -1, -1)
        self.policy_net.train()
        losses = []
        for i in range(n_steps):
            sl = slice(i * B, (i + 1) * B)
            q, sf, _ = self.policy_net(block_f[sl], binary[sl], action_f[sl], reward, obstacle)
            loss = self._loss(q, sf, q_target[sl], sf_target[sl] if use_sf else None)
            self.opt.zero_grad()
            if self._eager_reduce is None:
                self._eager_reduce = dqn_ops.ReduceTables(self.device)
            with dqn_ops.deferred_wgrad_reduce(self._eager_reduce):
                loss.backward()
            self.opt.step()
            losses.append(loss.detach())
        if defer:
            host = torch.empty(n_steps, dtype=torch.float32, pin_memory=True)
            host.copy_(torch.stack(losses), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            return DeferredLosses(host, done, None)
        return torch.stack(losses).tolist()

    # The replayed autograd step of the conv nets holds library reductions; one of that kind once returned garbage inside a
    # replayed graph (see _check_graph_losses).  The host learns of a bad loss one lock-step late (deferred read-back), so the
    # weights are protected ON THE DEVICE: parameters and Adam state are copied before the replays of a call and put back --
    # a torch.where on a device flag, no host decision -- when any of the call's losses is negative or not finite.  The
    # call's optimiser steps are then lost, not applied as garbage; the host switches to the eager step when it sees the loss.
    def _guard_tensors(self):
        flat = getattr(self.policy_net, "_flat_params", None)
        ts = [flat.flat] if flat is not None else [p.data for p in self.policy_net.parameters()]
        for s in self.opt.state.values():
            ts += [t for t in s.values() if torch.is_tensor(t) and t.is_cuda]
        adam = (self._graph_state or {}).get("adam")
        if adam is not None:
            ts.append(adam.step_count)
        return ts

    def _guard_snapshot(self, st):
        ts = self._guard_tensors()
        snap = st.get("guard")
        if snap is None or len(snap) != len(ts) or any(a.shape != b.shape for a, b in zip(snap, ts)):
            st["guard"] = [t.clone() for t in ts]
        else:
            torch._foreach_copy_(snap, ts)

    def _guard_restore(self, st, losses):
        bad = ~(torch.isfinite(losses).all() & (losses >= 0).all())
        for t, s in zip(self._guard_tensors(), st["guard"]):
            torch.where(bad, s, t, out=t)

    def _check_graph_losses(self, losses):
        """Guard for the anomaly recorded in DESIGN.md: a multi-workgroup reduction inside a replayed graph once returned
        garbage (a negative "MSE"; ROCm 7.2 + torch 2.10, cause not established).  The graph holds only single-workgroup
        reductions since, and every replay's loss is checked here: a sum of squares that is negative or not finite
        means a kernel in the graph misbehaved -- from then on the step runs eagerly."""
        if not all(np.isfinite(l) and l >= 0.0 for l in losses):
            warnings.warn(f"train-step graph produced an invalid loss {losses}; switching to the eager step")
            self.sync_optimizer_state()
            self._graph_state, self._eager_calls = None, -(1 << 30)
        return losses

    def sync_optimizer_state(self):
        """The captured step keeps Adam's step count itself (FusedSuccessorStep.adam_step); hand it back to the torch
        optimiser before its state is saved or ``optimizer.step()`` takes over again."""
        st = self._graph_state
        if st is not None and st.get("fused") and st["step"].fused_adam:
            st["step"].export_state()

    def train_step(self):
        out = self.train_steps(1)
        return out[0] if out else None

    def update_target(self):
        from robotoddler.training.successor_dqn import update_target_net
        update_target_net(self.policy_net, self.target_net, self.tau)

    # ------------------------------------------------------------------ checkpoint of what the nets / ring do not hold
    def save_extra(self, path, **counters):
        torch.save(dict(epsilon=float(self.epsilon), episodes_done=int(self.episodes_done), env_steps=int(self.env_steps),
                        step_images=self.step_images.cpu(), sample_gen=self.sample_gen.get_state().cpu(),
                        explore_gen=self.explore_gen.get_state().cpu(), counters={k: int(v) for k, v in counters.items()}), path)

    def load_extra(self, path):
        """Restores what save_extra wrote.  The file is rank 0's: exact continuation (same exploration draws, same
        env-step count) holds for a single-rank run.  With several ranks the shared items (epsilon, count images, replay
        sampling stream, episode counter) are restored everywhere, while a rank > 0 -- whose exploration stream and
        env-step count were never saved -- re-seeds its exploration stream from (seed, rank, lock-step) so that the
        ranks keep drawing different uniforms, and starts its local env-step count from rank 0's."""
        blob = torch.load(path, weights_only=True)
        self.epsilon, self.episodes_done, self.env_steps = blob["epsilon"], blob["episodes_done"], blob["env_steps"]
        self.step_images.copy_(blob["step_images"])
        self.sample_gen.set_state(blob["sample_gen"])
        if self.rank == 0:
            self.explore_gen.set_state(blob["explore_gen"])
        else:
            self.explore_gen.manual_seed(7654321 + self.seed * 1000 + self.rank + 7919 * (int(blob["counters"].get("lockstep", 0)) + 1))
        return blob["counters"]

    # ------------------------------------------------------------------ driver
    def lockstep(self, n_train_steps, defer_losses=False):
        """act -> all-gather -> replay push -> n optimiser steps -> soft update.  defer_losses=True returns the losses as
        a DeferredLosses (see train_steps): nothing after the replay push waits for the GPU, so the optimiser steps run
        under the host's queueing of the next lock-step."""
        rec, valid = self.act()
        if self.prioritized:
            rec[:, R.O_TD] = self.td_errors(rec).to(rec.dtype)
        # ONE wait per lock-step on this side: the two counts ride to pinned memory in front of the next act's candidate rows,
        # whose row count the host has to wait for anyway (bridges_valid_rows)
        done_rec = valid & (rec[:, R.O_DONE] > 0.5)
        self._counts_host.copy_(torch.stack([valid.sum(), done_rec.sum()]), non_blocking=True)
        arrived = torch.cuda.Event()
        arrived.record()
        self._rows(self.env, self._stable_flags(self.env))
        arrived.synchronize()                           # passed already unless the rows came out of the env's cache
        n_valid, n_done = int(self._counts_host[0]), int(self._counts_host[1])
        self.env_steps += n_valid
        allrec = D.all_gather_records(rec, valid, n_valid=n_valid)
        self.ring.push(allrec)
        if not D.active():
            self.episodes_done += n_done
        else:
            self.episodes_done += int((allrec[:, R.O_DONE] > 0.5).sum().item())
        losses = self.train_steps(n_train_steps, defer=defer_losses)
        self.update_target()
        self.epsilon = (self.epsilon - self.eps_end) * self.eps_decay + self.eps_end
        return losses, allrec


class DeferredLosses:
    """Losses of one train_steps call on their way to the host (pinned buffer + event)."""

    def __init__(self, host, done, agent):
        self._host, self._done, self._agent, self._list = host, done, agent, None

    def get(self):
        if self._list is None:
            if self._host is None:
                self._list = []
            else:
                self._done.synchronize()
                self._list = self._host.tolist()
                if self._agent is not None:
                    self._agent._check_graph_losses(self._list)
        return self._list


def lockstep_log_values(info):
    """What one lock-step hands to the aim / wandb sinks, under the reference's names (successor_dqn.py:489-499) where the
    quantity exists per lock-step: reward / lin_reward = mean over the lock-step's transitions (the reference: discounted
    sum over one episode), avg_loss, num_steps = env-steps of the lock-step on this rank, epsilon; plus the run counters."""
    return dict(reward=info['mean_reward'], lin_reward=info['mean_lin_reward'], avg_loss=info['avg_loss'],
                num_steps=info['lockstep_env_steps'], epsilon=info['epsilon'], env_steps=info['env_steps'],
                steps_per_s=info['steps_per_s'])


def run_vectorised(args, device, aim_run=None, wandb_run=None, return_agent=False):
    from robotoddler.training.successor_dqn import make_nets, track_run_sinks
    backend = os.environ.get("BRIDGES_DIST_BACKEND")          # 'gloo' = rehearsal with several ranks on one card
    rank, world = D.init(backend=backend, device=device)
    names = dict(trapezoid=["trapezoid"], hexagon=["hexagon"], both=["trapezoid", "hexagon"])[args['shapes']]
    geoms = [load_urdf(f"shapes/{n}.urdf") for n in names]
    if args.get('tower_height'):
        H, N = 0.8, args['tower_height']
        targets = [(0.5, 0, N * H + H / 2)]
        obstacles = [(0.5, 0., i * H + H / 2) for i in range(N)]
    else:
        sq, n = 0.6, args['bridge_length']
        targets = [(n * sq + 2.5 * sq, 0, sq / 2)]
        obstacles = [(i * sq, 0, sq / 2) for i in range(1, n + 1)]
    seed = args['seed'] or 0
    torch.manual_seed(seed)                                    # identical initial weights on every rank
    policy_net, target_net = make_nets(args, device)
    env = VecAssemblyGym(args['num_envs'], geoms, obstacles, targets, max_steps=args['max_steps'],
                         seed=seed * 1000003 + rank, device=device, env_id_base=rank * args['num_envs'],
                         f32_rasters=VecDQN.acting_needs_f32_rasters(policy_net),
                         img_size=args.get('image_size') or (64, 64))
    opt = torch.optim.Adam(policy_net.parameters(), lr=args['learning_rate'], fused=True)    # one launch for all tensors
    capacity = max(args['replay_buffer_capacity'], 4 * args['num_envs'] * world)
    agent = VecDQN(policy_net, target_net, opt, env, capacity, args['batch_size'], args['gamma'], args['tau'],
                   args['loss_function'], seed=seed, rank=rank, prioritized=args.get('prioritized_replay', False))
    history, t0, it = [], time.time(), 0
    next_ckpt = args['checkpoint_every']
    if args.get('load_checkpoint'):                      # successor_dqn.py:654-665 + utils.py:31-50 of the reference
        from robotoddler.utils.utils import load_checkpoint
        path = args['load_checkpoint']
        load_checkpoint(path, policy_net, target_net, agent.ring, opt,
                        devices=dict(policy_net=device, target_net=device, optimizer=device))
        it = agent.load_extra(os.path.join(path, 'agent.pt'))['lockstep']
        next_ckpt = (agent.episodes_done // args['checkpoint_every'] + 1) * args['checkpoint_every']
        env.reset()                                      # a checkpoint is taken with all environments freshly reset
    steps_at_start, t0 = agent.env_steps, time.time()    # throughput counts what THIS run (resumed or not) has stepped

    def finish(entry):
        """Fill in the numbers of a lock-step that were still on their way to the host when its entry was made."""
        info, deferred, stats_host, done = entry
        ls = deferred.get()
        info['avg_loss'] = float(np.mean(ls)) if ls else None
        if stats_host is not None:
            done.synchronize()
            info['mean_reward'], info['mean_lin_reward'] = float(stats_host[0]), float(stats_host[1])
        if rank == 0 and (aim_run is not None or wandb_run is not None):
            # one call per lock-step, step = episodes finished so far (the reference's x axis is the episode number)
            track_run_sinks(lockstep_log_values(info), info['episodes'], 'training', aim_run=aim_run, wandb_run=wandb_run)
        if args['verbose'] and rank == 0:
            print(info)

    pending = None
    while agent.episodes_done < args['num_episodes']:
        # losses and record statistics are read one lock-step late: nothing here waits for the optimiser steps, so they
        # run while the host queues the next lock-step's acting
        steps_before = agent.env_steps
        losses, rec = agent.lockstep(args['num_training_steps'], defer_losses=True)
        it += 1
        if args.get('save_checkpoint') and agent.episodes_done >= next_ckpt:                  # utils.py:54-89 layout
            agent.sync_optimizer_state()
            if rank == 0:
                from robotoddler.utils.utils import save_checkpoint
                save_checkpoint(args['save_checkpoint'], policy_net, target_net, agent.ring, opt, agent.episodes_done,
                                {k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(v)) for k, v in args.items()})
                agent.save_extra(os.path.join(args['save_checkpoint'], str(agent.episodes_done), 'agent.pt'), lockstep=it)
            next_ckpt = (agent.episodes_done // args['checkpoint_every'] + 1) * args['checkpoint_every']
            # the single-env reference checkpoints between episodes; the lock-step analogue: every rank starts all its
            # environments afresh, so that a resumed run (fresh environments) continues exactly like this one
            env.reset()
        if it % 100 == 0:
            D.broadcast_module(policy_net)
            D.broadcast_module(target_net)
        stats_host, done = None, None
        if rec.numel():
            stats_host = torch.empty(2, dtype=torch.float64, pin_memory=True)
            stats_host.copy_(torch.stack([rec[:, R.O_REWARD].mean(), rec[:, R.O_LIN].mean()]).double(), non_blocking=True)
            done = torch.cuda.Event()
            done.record()
        info = dict(lockstep=it, episodes=agent.episodes_done, env_steps=agent.env_steps, avg_loss=None, mean_reward=None,
                    mean_lin_reward=None, epsilon=agent.epsilon, lockstep_env_steps=agent.env_steps - steps_before,
                    steps_per_s=(agent.env_steps - steps_at_start) * world / max(time.time() - t0, 1e-9))
        history.append(info)
        if pending is not None:
            finish(pending)
        pending = (info, losses, stats_host, done)
    if pending is not None:
        finish(pending)
    agent.sync_optimizer_state()       # the captured step counts Adam's steps itself: hand the count back before anyone reads opt.state
    return (history, agent) if return_agent else history
