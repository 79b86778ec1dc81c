"""Replay buffers (robotoddler/utils/replay_memory.py:10-93 of the reference): a deque of Transition namedtuples,
uniform ``random.sample``; ``stack_tensors`` concatenates every tensor field along dim 0 (the next-* fields are
ragged: one row per next action)."""
import random
from collections import deque

import numpy as np
import torch


def _plain(x):
    """Checkpoint form of one Transition field: tensors and numbers stay, Action dataclasses (and lists of them) become
    plain dicts -- so the file holds nothing but tensors and containers and loads with torch.load(weights_only=True)."""
    import dataclasses
    if dataclasses.is_dataclass(x) and not isinstance(x, type):
        return {"__action__": dataclasses.asdict(x)}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    return x


def _unplain(x, action_cls):
    if isinstance(x, dict) and "__action__" in x:
        return action_cls(**x["__action__"])
    if isinstance(x, list):
        return [_unplain(v, action_cls) for v in x]
    return x


def _save_plain(path, memory, extra=None):
    items = [[_plain(v) for v in t] for t in memory]
    fields = list(memory[0]._fields) if len(memory) else []
    torch.save(dict(format=2, fields=fields, items=items, maxlen=getattr(memory, "maxlen", None), extra=extra), path)


def _load_plain(path):
    """-> (list of Transition, maxlen, extra).  Refuses anything that is not the plain format above: pickled deques of
    the reference's own save() would need a full unpickle, which can run code from the file."""
    blob = torch.load(path, weights_only=True)
    if not (isinstance(blob, dict) and blob.get("format") == 2):
        raise ValueError(f"{path}: not a replay buffer saved by this build (plain-tensor format 2)")
    from assembly_gym.envs.gym_env import Action
    from robotoddler.training.successor_dqn import Transition
    if blob["items"] and list(blob["fields"]) != list(Transition._fields):
        raise ValueError(f"{path}: Transition fields differ")
    return [Transition(*[_unplain(v, Action) for v in t]) for t in blob["items"]], blob["maxlen"], blob.get("extra")


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
        """Same role as replay_memory.py:33-36 of the reference (which pickles the deque); here as plain tensors."""
        _save_plain(path, self.memory)

    def load(self, path):
        items, maxlen, _ = _load_plain(path)
        self.memory = items if maxlen is None else deque(items, maxlen=maxlen)

    def draw(self, batch_size=None):
        """The transitions of one sample() call, unstacked (the same use of the random generator)."""
        return self.memory if batch_size is None else random.sample(self.memory, batch_size)

    def sample(self, batch_size=None, stack_tensors=False, device=None):
        batch = self.draw(batch_size)
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
        _save_plain(path, self.memory, extra=[float(p) for p in self.priorities])

    def load(self, path):
        items, maxlen, pr = _load_plain(path)
        self.memory = items if maxlen is None else deque(items, maxlen=maxlen)
        self.priorities = list(pr) if maxlen is None else deque(pr, maxlen=maxlen)

    def draw(self, batch_size=None):
        if batch_size is None:
            return self.memory
        pr = [p.cpu().item() if isinstance(p, torch.Tensor) else p for p in self.priorities]
        total = sum(pr)
        probs = [1 / len(pr)] * len(pr) if total < 1e-10 else [p / total for p in pr]
        idx = np.random.choice(len(self.memory), batch_size, p=probs)
        return [self.memory[i] for i in idx]

    def sample(self, batch_size=None, stack_tensors=False, device=None):
        batch = self.draw(batch_size)
        if stack_tensors:
            return batch, _stack(batch, device)
        return batch, batch[0].__class__(*zip(*batch))
