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


class EpsilonGreedyOracle:
    """successor_dqn.py:98-132 statement by statement (plain torch on the CPU).  ``rand`` is the uniform source
    (``random.random`` in the reference), injected so a test can script the explore / exploit decisions."""

    def __init__(self, eps_start=0.5, eps_end=0.05, gamma=0.99, episode=0, max_steps=10, rand=None):
        import random
        self.epsilon = (eps_start - eps_end) * (gamma ** episode) + eps_end
        self.eps_end, self.gamma = eps_end, gamma
        self.step_images = [torch.zeros(64, 64) for _ in range(max_steps)]
        self.rand = rand or random.random

    def step(self):                                          # :107-109
        self.epsilon = (self.epsilon - self.eps_end) * self.gamma + self.eps_end
        return self

    def __call__(self, q_values, step_index, action_features):
        if self.rand() > self.epsilon:                       # :112-113
            return torch.argmax(q_values).item()
        sel, new_image = explore_choice(self.step_images[step_index], action_features)      # :116-129
        self.step_images[step_index] = new_image
        return sel


def vectorised_explore(step_images, step_of_env, rasters_of_env, explore):
    """The batched reading of the exploration rule used by the vectorised loop (vec_dqn.VecDQN.act): every exploring
    env applies successor_dqn.py:116-126 against the count images AS THEY WERE AT THE START of the lock-step (envs of
    one lock-step do not see each other's choices), and the chosen rasters are added afterwards (:129), env by env.

    step_images [S,64,64] float; step_of_env[e] = episode step of env e; rasters_of_env[e] = [A_e,64,64] float rasters
    of its valid candidates (may be empty); explore[e] bool.  Returns (selection per env or None, new step_images)."""
    start = step_images.clone()
    out = step_images.clone()
    sel = []
    for e, feats in enumerate(rasters_of_env):
        if not explore[e] or len(feats) == 0:
            sel.append(None)
            continue
        s, _ = explore_choice(start[step_of_env[e]], feats.unsqueeze(1))
        sel.append(s)
        out[step_of_env[e]] += feats[s]
    return sel, out


def log_episode_values(rewards, lin_rewards, losses, gamma, last_next_binary, epsilon=None):
    """successor_dqn.py:484-503: the numbers log_episode reports for one episode.  rewards / lin_rewards: per-step
    floats in order; last_next_binary: the 6 binary features of the state the last transition led to."""
    info = {
        'reward': sum(gamma ** i * r for i, r in enumerate(rewards)),
        'lin_reward': sum(gamma ** i * r for i, r in enumerate(lin_rewards)),
        'avg_loss': sum(losses) / len(losses) if losses else None,
        'num_steps': len(rewards),
        'stable': float(last_next_binary[0]),
        'collision': float(last_next_binary[1]),
    }
    if epsilon is not None:
        info['epsilon'] = epsilon
    return info
