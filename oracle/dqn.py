"""Oracle: line-by-line restatement (plain torch on the CPU, float32 as the reference computes) of the parts of
robotoddler/training/successor_dqn.py that cannot be imported (the module needs aim / wandb / compas):

* target construction + losses of ``train_policy_net`` (successor_dqn.py:176-234),
* ``update_target_net`` (successor_dqn.py:280-288),
* the count-based exploration rule of ``EpsilonGreedy.__call__`` (successor_dqn.py:112-132).

The Q-networks themselves ARE importable from the reference; ``tests/golden/make_net_fixtures.py`` runs them to
produce the pinned inputs/outputs these functions are checked with.
"""
import numpy as np
import torch


def td_targets(next_q_values, next_succ_block_features, num_actions, done, gamma, lin_reward, action_features):
    """successor_dqn.py:197-213 + the targets of :222 and :230.

    Returns (q_target as the reference broadcasts it, next_q selected+masked [B], state_target [B,H,W] or None,
    selected row indices)."""
    offsets = np.cumsum([0] + list(num_actions))
    selected = [chunk.argmax().item() + off for chunk, off in zip(next_q_values.split(list(num_actions)), offsets)]
    nq = next_q_values[selected].clone()
    done_mask = torch.tensor(done, dtype=bool)
    nq[done_mask] = 0
    q_target = lin_reward + gamma * nq                      # reference shapes: [B,1] + [B] -> [B,B]
    state_target = None
    if next_succ_block_features is not None:
        nsf = next_succ_block_features[selected][:, 0].clone()
        nsf[done_mask] = 0
        state_target = action_features.squeeze(1) + gamma * nsf
    return q_target, nq, state_target, selected


def losses(q_values, succ_block_features, q_target, state_target, loss_fct):
    """successor_dqn.py:216-234."""
    mse = torch.nn.MSELoss()
    loss = 0.
    parts = loss_fct.split('+')
    if 'mse_q_values' in parts:
        loss = loss + mse(q_values, q_target)
    if 'mse_block_features' in parts:
        loss = loss + mse(succ_block_features[:, 0], state_target)
    return loss


def update_target_net(policy_sd, target_sd, tau):
    """successor_dqn.py:280-288 on state_dicts; returns the new target state_dict."""
    return {k: policy_sd[k] * tau + target_sd[k] * (1 - tau) for k in policy_sd}


def explore_choice(step_image, action_features):
    """successor_dqn.py:116-129: index of the least-overlapping action and the updated step image."""
    scores = [torch.sum(step_image * a.squeeze(0)).item() for a in action_features]
    sel = int(torch.argmin(torch.tensor(scores)).item())
    return sel, step_image + action_features[sel].squeeze(0)
