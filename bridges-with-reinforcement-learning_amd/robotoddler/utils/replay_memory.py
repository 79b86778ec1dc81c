"""Replay buffers (robotoddler/utils/replay_memory.py:10-93 of the reference): a deque of Transition namedtuples,
uniform ``random.sample``; ``stack_tensors`` concatenates every tensor field along dim 0 (the next-* fields are
ragged: one row per next action)."""
import random
from collections import deque

import numpy as np
import torch


def _stack(batch, device):
    cls = batch[0].__class__
    return cls(*[torch.cat(x).to(device=device) if torch.is_tensor(x[0]) else x for x in zip(*batch)])


class ReplayBuffer:
    def __init__(self, capacity=None):
        self.memory = [] if capacity is None else deque([], maxlen=capacity)

    def push(self, transition):
        for t in (transition if isinstance(transition, list) else [transition]):
            self.memory.append(t)

    def save(self, path):
        torch.save(self.memory, path)

    def load(self, path):
        self.memory = torch.load(path, weights_only=False)

    def sample(self, batch_size=None, stack_tensors=False, device=None):
        batch = self.memory if batch_size is None else random.sample(self.memory, batch_size)
        if stack_tensors:
            return batch, _stack(batch, device)
        return batch, batch[0].__class__(*zip(*batch))

    def __len__(self):
        return len(self.memory)


class PrioritizedReplayBuffer(ReplayBuffer):
    """Priority = |td_error| + 1e-5, sampled with numpy's global RNG (replay_memory.py:45-93)."""

    def __init__(self, capacity=None, gamma=0.99, policy_net=None, target_net=None, device=None):
        super().__init__(capacity)
        self.priorities = [] if capacity is None else deque([], maxlen=capacity)
        self.gamma, self.policy_net, self.target_net, self.device = gamma, policy_net, target_net, device

    def push(self, transition):
        for t in (transition if isinstance(transition, list) else [transition]):
            self.memory.append(t)
            self.priorities.append(t.td_error + 1e-5)

    def save(self, path):
        torch.save((self.memory, self.priorities), path)

    def load(self, path):
        self.memory, self.priorities = torch.load(path, weights_only=False)

    def sample(self, batch_size=None, stack_tensors=False, device=None):
        if batch_size is None:
            batch = self.memory
        else:
            pr = [p.cpu().item() if isinstance(p, torch.Tensor) else p for p in self.priorities]
            total = sum(pr)
            probs = [1 / len(pr)] * len(pr) if total < 1e-10 else [p / total for p in pr]
            idx = np.random.choice(len(self.memory), batch_size, p=probs)
            batch = [self.memory[i] for i in idx]
        if stack_tensors:
            return batch, _stack(batch, device)
        return batch, batch[0].__class__(*zip(*batch))
