"""Per-layer inference time of the conv Q-networks' 3x3 layers at an acting-sized batch: the library convolution + the fused
bias / ReLU pass (what _conv_relu runs) against bridges_conv3x3 with the bias / ReLU epilogue (the training kernel, k_c3).
  python tools/conv_layer_bench.py [--rows 2048]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import torch
import torch.nn.functional as F
from bridges_hip import dqn_ops
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=2048)
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
LAYERS = [("first layer", 4, 16, 64), ("unet e21", 16, 32, 32), ("unet e22 / d32", 32, 32, 32), ("unet e31", 32, 64, 16), ("unet e32", 64, 64, 16), ("unet d31", 64, 32, 32),
          ("unet d41", 32, 16, 64), ("unet e12 / d42", 16, 16, 64), ("convnet 1b", 16, 16, 64), ("convnet 2a", 16, 32, 32),
          ("convnet 3a", 32, 64, 16), ("convnet 3b", 64, 64, 16), ("convnet 4a", 64, 128, 8), ("convnet 4b", 128, 128, 8)]


def timed(f, reps=5):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"{'layer':18s} {'c_in':>4s} {'c_out':>5s} {'W':>3s} {'library us':>11s} {'k_c3 us':>9s} {'TFLOP/s':>8s}  max|diff|  [o16 inference kernel us, plain / pooled]")
with torch.no_grad():
    for name, ci, co, W in LAYERS:
        x = torch.relu(torch.randn(a.rows, ci, W, W, device=dev))
        w = torch.randn(co, ci, 3, 3, device=dev) * (2.0 / (9 * ci)) ** 0.5
        b = torch.randn(co, device=dev) * 0.1
        lib = lambda: dqn_ops.bias_relu_(F.conv2d(x, w, None, 1, 1).contiguous(), b)
        own = lambda: dqn_ops.conv3x3(x, w, b)
        t_lib, t_own = timed(lib), timed(own)
        diff = float((lib() - own()).abs().max())
        flops = 2.0 * a.rows * W * W * ci * co * 9
        extra = ""
        if co == 16 and W == 64 and ci in (4, 16):
            extra = f"  [{timed(lambda: dqn_ops.conv3x3_relu_o16(x, w, b)):.1f} / {timed(lambda: dqn_ops.conv3x3_relu_o16(x, w, b, pool=True)):.1f}]"
        print(f"{name:18s} {ci:4d} {co:5d} {W:3d} {t_lib:11.1f} {t_own:9.1f} {flops / t_own / 1e6:8.1f}  {diff:.2e}{extra}", flush=True)
