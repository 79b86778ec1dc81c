"""Generates tests/golden/net_fixtures.pt by RUNNING the importable reference modules
(robotoddler.models.cv, robotoddler.utils.utils, robotoddler.utils.replay_memory -- torch/numpy only) on seeded
inputs.  Run in the build container:  python tests/golden/make_net_fixtures.py
The file holds tensors only (inputs, state_dicts, outputs); it is loaded with torch.load(weights_only=True).
"""
import os
import random
import sys
import warnings

import torch

REF = "/root/reference"
sys.path.insert(0, REF)
from robotoddler.models.cv import ConvNet, Policy, SuccessorMLP, UNet          # noqa: E402  (reference code)
from robotoddler.utils.replay_memory import ReplayBuffer                        # noqa: E402
from robotoddler.utils.utils import convolve_with_gaussian, gaussian_kernel, init_weights   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
out = {}


def inputs(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: (torch.rand(*s, generator=g) > 0.7).float()
    return dict(block=r(n, 1, *size), binary=(torch.rand(n, 6, generator=g) > 0.5).float(), action=r(n, 1, *size),
                reward=torch.rand(n, 1, *size, generator=g), obstacle=r(n, 1, *size))


def args5(i):
    return i["block"], i["binary"], i["action"], i["reward"], i["obstacle"]


def half_state(net):
    """Store weights as fp16 (small fixture); the net is reloaded with the rounded values before any output is taken."""
    st = {k: v.half() for k, v in net.state_dict().items()}
    net.load_state_dict({k: v.float() for k, v in st.items()})
    return st


# --- SuccessorMLP on 16x16 images (size-generic code path, small fixture) ---
torch.manual_seed(1)
net = SuccessorMLP(img_size=(16, 16), hidden_dims=[32, 16, 32])
net.apply(init_weights)
i = inputs(5, (16, 16), 11)
q, sf, sb = net(*args5(i))
out["successor_mlp"] = dict(state=net.state_dict(), inputs=i, q=q.detach(), sf=sf.detach(), sb=sb.detach())

# --- ConvNet: 2-argument forward as at HEAD, and the 5-argument forward of cv.py:67-73 composed from its own
#     submodules (the commented block, executed statement by statement) ---
torch.manual_seed(2)
net = ConvNet(in_channels=2, img_size=(16, 16), num_features=0)
net.apply(init_weights)
i = inputs(4, (16, 16), 12)
st = half_state(net)
out["convnet2"] = dict(state=st, inputs=i, out=net(i["block"], i["action"]).detach())
torch.manual_seed(3)
net = ConvNet(in_channels=4, img_size=(16, 16), num_features=6)
net.apply(init_weights)
i = inputs(4, (16, 16), 13)
st = half_state(net)
x = torch.cat([i["block"], i["action"], i["reward"], i["obstacle"]], dim=1)
x = net.layers(x)
x = net.mlp(torch.cat([x.view(-1, net.bottleneck_size), i["binary"]], dim=1))
out["convnet5"] = dict(state=st, inputs=i, q=x[:, 0].detach(), sb=x[:, 1:].view(-1, 2, 6).detach())

# --- UNet(1) / UNet(2) on 16x16, Policy on 64x64 (its ConvNet head is hard-wired to 64x64) ---
for n_class in (1, 2):
    torch.manual_seed(4 + n_class)
    net = UNet(n_class)
    net.apply(init_weights)
    i = inputs(3, (16, 16), 14 + n_class)
    st = half_state(net)
    out[f"unet{n_class}"] = dict(state=st, inputs=i, out=net(*args5(i)).detach())
torch.manual_seed(7)
net = Policy()
net.apply(init_weights)
i = inputs(2, (64, 64), 17)
i = {k: v.half().float() for k, v in i.items()}
st = half_state(net)
q, sf, stab = net(*args5(i))
out["policy"] = dict(state=st, inputs={k: v.half() for k, v in i.items()}, q=q.detach(), sf=sf.detach(), stability=stab.detach())

# --- parameter counts quoted in SURVEY.md §8 a-17 ---
out["param_counts"] = dict(
    successor_mlp=sum(p.numel() for p in SuccessorMLP(img_size=(64, 64), hidden_dims=[256, 128, 64, 128, 256]).parameters()),
    convnet=sum(p.numel() for p in ConvNet(img_size=(64, 64)).parameters()),
    policy=sum(p.numel() for p in Policy().parameters()))

# --- init_weights under a seed: checksums of every tensor ---
torch.manual_seed(21)
net = SuccessorMLP(img_size=(8, 8), hidden_dims=[8])
net.apply(init_weights)
out["init_weights"] = {k: v.clone() for k, v in net.state_dict().items()}

# --- gaussian target map ---
out["gaussian_kernel_101_16"] = gaussian_kernel(101, 16)
img = torch.zeros(64, 64)
img[40:44, 20:24] = 1.0
out["gaussian_conv"] = dict(inp=img, out=convolve_with_gaussian(img, 101, 16))

# --- ReplayBuffer: deque capacity + random.sample order ---
rb = ReplayBuffer(capacity=5)
from collections import namedtuple
T = namedtuple("T", "a b")
rb.push([T(torch.tensor([float(k)]), k) for k in range(8)])
random.seed(5)
batch, stacked = rb.sample(batch_size=3, stack_tensors=True)
out["replay"] = dict(kept=torch.tensor([t.b for t in rb.memory]), sampled=torch.tensor([t.b for t in batch]), stacked_a=stacked.a)

torch.save(out, os.path.join(HERE, "net_fixtures.pt"))
print({k: (list(v.keys()) if isinstance(v, dict) else tuple(v.shape)) for k, v in out.items()})
print("size", os.path.getsize(os.path.join(HERE, "net_fixtures.pt")))
