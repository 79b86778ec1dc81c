"""Host time of one library GEMM call (torch._addmm_activation -> hipBLASLt) for a row count seen before / never seen.
Usage: python tools/addmm_host_cost.py"""
import time
import torch

dev = torch.device("cuda:0")
W = torch.randn(128, 256, device=dev)
b = torch.randn(128, device=dev)
x = torch.randn(60000, 256, device=dev)
torch._addmm_activation(b, x[:1000], W.T)
torch.cuda.synchronize()
for label, ns in (("same n", [45000] * 40), ("new n every call", list(range(44000, 44040))), ("same n again", [45000] * 40),
                  ("multiples of 512", [45056 + 512 * (i % 3) for i in range(40)])):
    ts = []
    for n in ns:
        torch.cuda.synchronize()
        t = time.perf_counter()
        torch._addmm_activation(b, x[:n], W.T)
        ts.append(time.perf_counter() - t)
    torch.cuda.synchronize()
    ts.sort()
    print(f"{label:20s} median host time per call {ts[len(ts) // 2] * 1e6:7.1f} us   max {ts[-1] * 1e6:8.1f} us")
