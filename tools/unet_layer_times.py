"""Per-operation times of the U-Net policy's inference forward (robotoddler/models/cv.py UNet) on 2048 rows: which of
its ~20 operations the 8 ms go to.  Usage: python tools/unet_layer_times.py"""
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd"))
from robotoddler.models.cv import UNet, _conv_relu      # noqa: E402
from robotoddler.utils.utils import init_weights        # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
n = 2048
net = UNet(1).to(dev)
net.apply(init_weights)
net.eval()
x = torch.rand(n, 4, 64, 64, device=dev)


def timed(name, fn, reps=8):
    with torch.no_grad():
        for _ in range(3):
            out = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = fn()
        torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name:28s} {ms:7.3f} ms", flush=True)
    return out, ms


total = 0.0
with torch.no_grad():
    a, t = timed("e11 4->16 @64", lambda: _conv_relu(net.e11, x)); total += t
    s1, t = timed("e12 16->16 @64", lambda: _conv_relu(net.e12, a)); total += t
    p1, t = timed("pool1", lambda: net.pool1(s1)); total += t
    a, t = timed("e21 16->32 @32", lambda: _conv_relu(net.e21, p1)); total += t
    s2, t = timed("e22 32->32 @32", lambda: _conv_relu(net.e22, a)); total += t
    p2, t = timed("pool2", lambda: net.pool2(s2)); total += t
    a, t = timed("e31 32->64 @16", lambda: _conv_relu(net.e31, p2)); total += t
    b, t = timed("e32 64->64 @16", lambda: _conv_relu(net.e32, a)); total += t
    up, t = timed("upconv3 64->32 (x2)", lambda: net.upconv3(b)); total += t
    u, t = timed("cat(up3, s2)", lambda: torch.cat([up, s2], dim=1)); total += t
    a, t = timed("d31 64->32 @32", lambda: _conv_relu(net.d31, u)); total += t
    a, t = timed("d32 32->32 @32", lambda: _conv_relu(net.d32, a)); total += t
    up4, t = timed("upconv4 32->16 (x2)", lambda: net.upconv4(a)); total += t
    u, t = timed("cat(up4, s1)", lambda: torch.cat([up4, s1], dim=1)); total += t
    a, t = timed("d41 32->16 @64", lambda: _conv_relu(net.d41, u)); total += t
    a, t = timed("d42 16->16 @64", lambda: _conv_relu(net.d42, a)); total += t
    o, t = timed("outconv 1x1 16->1", lambda: net.outconv(a)); total += t
    print(f"{'sum':28s} {total:7.3f} ms")
    inputs = [x[:, :1], torch.zeros(n, 6, device=dev), x[:, 1:2], x[:, 2:3], x[:, 3:4]]
    timed("UNet forward (whole)", lambda: net(*inputs))
