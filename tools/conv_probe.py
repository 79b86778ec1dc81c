"""Timing probe for bridges_conv3x3_relu_o16 (16 -> 16 channels, 2048 images of 64x64): input statistics (uniform, post-ReLU,
zeros, large) and burst length do not move it -- what did was a guarded weight load that the compiler kept behind the LDS
staging (0.65 against 0.43 ms).  Usage: python tools/conv_probe.py"""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bridges-with-reinforcement-learning_amd"))
from bridges_hip import dqn_ops
dev = torch.device("cuda")
torch.manual_seed(0)
def t(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
w = torch.randn(16, 16, 3, 3, device=dev) * 0.1
b = torch.randn(16, device=dev) * 0.1
for name, x in (("rand", torch.rand(2048, 16, 64, 64, device=dev)), ("relu(randn)", torch.relu(torch.randn(2048, 16, 64, 64, device=dev))),
                ("zeros", torch.zeros(2048, 16, 64, 64, device=dev)), ("randn*100", torch.randn(2048, 16, 64, 64, device=dev) * 100)):
    for reps in (10, 200):
        print(name, reps, "reps: %.3f ms" % t(lambda: dqn_ops.conv3x3_relu_o16(x, w, b, False), reps), flush=True)
out = torch.empty(2048, 16, 64, 64, device=dev)
x = torch.rand(2048, 16, 64, 64, device=dev)
import ctypes as C
from bridges_hip import abi
L = abi.lib()
def raw():
    L.bridges_conv3x3_relu_o16(C.c_void_p(x.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), 2048, 16, 64, 64, 0, C.c_void_p(torch.cuda.current_stream().cuda_stream))
print("fixed output buffer, 200 reps: %.3f ms" % t(raw, 200))
