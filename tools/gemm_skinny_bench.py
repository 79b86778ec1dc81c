"""Times the three large GEMMs of one SuccessorMLP optimiser step at batch 32 (first layer forward / weight gradient,
last layer forward / weight gradient / input gradient) under the BLAS back-ends torch offers, against the time their
weight traffic takes at HBM speed.  Usage: python tools/gemm_skinny_bench.py [--tunable]"""
import os
import sys
import time

if "--tunable" in sys.argv:
    os.environ["PYTORCH_TUNABLEOP_ENABLED"] = "1"
    os.environ.setdefault("PYTORCH_TUNABLEOP_FILENAME", "/tmp/tunableop_%d.csv")
import torch


def t(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device("cuda:0")
    B = 32
    shapes = [("layer1", 4 * 4096 + 6, 256), ("head", 256, 2 * 4096 + 12)]
    for lib in ("hipblaslt", "hipblas"):
        try:
            torch.backends.cuda.preferred_blas_library(lib)
        except Exception as exc:
            print(lib, "unavailable", exc)
            continue
        for name, K, N in shapes:
            x = torch.randn(B, K, device=dev)
            w = torch.randn(N, K, device=dev)
            bias = torch.randn(N, device=dev)
            dy = torch.randn(B, N, device=dev)
            fwd = t(lambda: torch.nn.functional.linear(x, w, bias))
            dw = t(lambda: dy.t() @ x)
            dx = t(lambda: dy @ w)
            floor = N * K * 4 / 5.0e12 * 1e6
            print(f"{lib:10s} {name:7s} K={K:6d} N={N:5d}: fwd {fwd:6.1f} us  dW {dw:6.1f} us  dx {dx:6.1f} us   (weights at 5 TB/s: {floor:4.1f} us)",
                  flush=True)


if __name__ == "__main__":
    main()
