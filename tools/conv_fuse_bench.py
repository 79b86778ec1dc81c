"""Per-layer timing of the ConvNet's conv3x3 + bias + ReLU on 2048 rows: NCHW vs channels_last tensors (MIOpen picks
NHWC implicit-GEMM kernels either way; with NCHW tensors it transposes around them)."""
import torch, time
import torch.nn.functional as F
dev = torch.device("cuda")
torch.manual_seed(0)
for (cin, cout, hw) in ((4, 16, 64), (16, 16, 64), (32, 16, 64), (16, 32, 32), (32, 32, 32), (32, 64, 16), (64, 64, 16), (64, 128, 8), (128, 128, 8)):
    x = torch.rand(2048, cin, hw, hw, device=dev)
    w = torch.randn(cout, cin, 3, 3, device=dev) * 0.1
    b = torch.randn(cout, device=dev) * 0.1
    xc, wc = x.contiguous(memory_format=torch.channels_last), w.contiguous(memory_format=torch.channels_last)
    res = {}
    with torch.no_grad():
        for name, fn in (("nchw conv+b+relu", lambda: F.relu(F.conv2d(x, w, b, padding=1))),
                         ("nchw conv only", lambda: F.conv2d(x, w, None, padding=1)),
                         ("nhwc conv+b+relu", lambda: F.relu(F.conv2d(xc, wc, b, padding=1))),
                         ("nhwc conv only", lambda: F.conv2d(xc, wc, None, padding=1)),
                         ("nhwc conv+b+relu+pool", lambda: F.max_pool2d(F.relu(F.conv2d(xc, wc, b, padding=1)), 2)),
                         ("nchw conv+b+relu+pool", lambda: F.max_pool2d(F.relu(F.conv2d(x, w, b, padding=1)), 2))):
            for _ in range(3): o = fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): o = fn()
            torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 10 * 1e3
    if cout == 16 and hw == 64:
        import os, sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bridges-with-reinforcement-learning_amd"))
        from bridges_hip import dqn_ops
        for name, fn in (("HAND-WRITTEN conv+b+relu", lambda: dqn_ops.conv3x3_relu_o16(x, w, b, False)),
                         ("HAND-WRITTEN conv+b+relu+pool", lambda: dqn_ops.conv3x3_relu_o16(x, w, b, True))):
            for _ in range(3): o = fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(10): o = fn()
            torch.cuda.synchronize(); res[name] = (time.perf_counter() - t0) / 10 * 1e3
    gf = 2048 * hw * hw * cout * 9 * cin * 2 / 1e9
    print(f"{cin}->{cout}@{hw} ({gf:.1f} GFLOP): " + "; ".join(f"{k} {v:.3f} ms" for k, v in res.items()), flush=True)
