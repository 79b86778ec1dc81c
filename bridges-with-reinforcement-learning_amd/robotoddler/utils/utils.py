"""Training utilities (robotoddler/utils/utils.py:12-115 of the reference): weight init, image-size parsing,
checkpoint directory layout (<path>/<episode>/{policy_net,target_net,optimizer,replay_buffer}.pt + meta.json +
'latest' symlink), Gaussian target map."""
import argparse
import json
import os
from datetime import datetime

import torch
import torch.nn.functional as F
from torch import nn


def init_weights(m):
    """xavier_uniform weights, bias 0.01 for Conv2d / Linear (utils.py:12-19)."""
    if type(m) in (nn.Conv2d, nn.Linear):
        torch.nn.init.xavier_uniform_(m.weight)
        m.bias.data.fill_(0.01)


def parse_img_size(s):
    try:
        return tuple(map(int, s.split('x')))
    except Exception:
        raise argparse.ArgumentTypeError("Image size must be a tuple of integers.")


def optimizer_to(optimizer, device):
    """Move the optimiser's per-parameter state tensors (utils.py:21-28 of the reference)."""
    for per_param in optimizer.state.values():
        per_param.update({name: t.to(device) for name, t in per_param.items() if isinstance(t, torch.Tensor)})


def save_checkpoint(path, policy_net, target_net, replay_buffer, optimizer, episode, config, aim_run=None, wandb_run=None):
    current = os.path.join(path, str(episode))                  # <path>/<episode>/..., utils.py:54-89 of the reference
    os.makedirs(current, exist_ok=True)
    for stem, holder in (("policy_net", policy_net), ("target_net", target_net), ("optimizer", optimizer)):
        torch.save(holder.state_dict(), os.path.join(current, stem + ".pt"))
    replay_buffer.save(os.path.join(current, "replay_buffer.pt"))
    meta = dict(episode=episode, timestamp=str(datetime.now()), config=config)
    if aim_run is not None:
        meta['aim_hash'] = aim_run.hash
    with open(os.path.join(current, 'meta.json'), 'w') as f:
        json.dump(meta, f, indent=2, default=str)
    latest = os.path.join(path, 'latest')
    if os.path.lexists(latest):
        os.remove(latest)
    os.symlink(os.path.abspath(current), latest)


def load_checkpoint(path, policy_net, target_net, replay_buffer, optimizer, devices=None):
    if not os.path.exists(path) or not os.path.isdir(path):
        raise FileNotFoundError(f"Path {path} does not exist or is not a directory.")
    with open(os.path.join(path, 'meta.json')) as f:
        meta = json.load(f)
    devices = devices or dict()
    ld = lambda stem: torch.load(os.path.join(path, stem + '.pt'), map_location=devices.get(stem), weights_only=True)
    policy_net.load_state_dict(ld('policy_net'))
    target_net.load_state_dict(ld('target_net'))
    optimizer.load_state_dict(ld('optimizer'))
    replay_buffer.load(os.path.join(path, 'replay_buffer.pt'))
    return meta


def gaussian_kernel(kernel_size, sigma):
    coords = torch.arange(kernel_size) - kernel_size // 2
    k = torch.exp(-(coords.float() ** 2) / (2 * sigma ** 2))
    k = k / k.sum()
    return k.unsqueeze(0) * k.unsqueeze(1)


_GAUSSIAN_CACHE = {}


def convolve_with_gaussian(input_tensor, kernel_size, sigma):
    """utils.py:107-115 of the reference.  For a tensor on the GPU the sum is evaluated on the host in one fixed order
    (bridges_hip.vec_env.gaussian_reward_map): the library convolution's algorithm -- hence the last bits of the reward map and
    everything trained on it -- depended on MIOpen's find results of the box.  Host tensors take the reference's own call."""
    if input_tensor.is_cuda:
        from bridges_hip.vec_env import gaussian_reward_map
        img = input_tensor.detach().cpu().numpy()
        key = (img.shape, img.dtype.str, img.tobytes(), kernel_size, sigma)    # the targets of a task are fixed: one evaluation per task
        out = _GAUSSIAN_CACHE.get(key)
        if out is None:
            if len(_GAUSSIAN_CACHE) >= 16:
                _GAUSSIAN_CACHE.clear()
            out = _GAUSSIAN_CACHE[key] = gaussian_reward_map(img, kernel_size, sigma)
        return torch.from_numpy(out).to(device=input_tensor.device, dtype=input_tensor.dtype)
    kernel = gaussian_kernel(kernel_size, sigma).to(input_tensor.device)
    return F.conv2d(input_tensor[None, None], kernel[None, None], padding=kernel_size // 2)[0, 0]
