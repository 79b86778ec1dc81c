"""Calibrates the attainable HBM write rate on this box: torch fill of a raster-sized buffer."""
import time, torch
n = 4_400_000_000 // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(3): x.zero_()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): x.zero_()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"fill {n*4/1e9:.2f} GB in {dt*1e3:.3f} ms = {n*4/dt/1e12:.2f} TB/s")
y = torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): y.copy_(x)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"copy {n*4/1e9:.2f} GB in {dt*1e3:.3f} ms = {2*n*4/dt/1e12:.2f} TB/s (read+write)")
