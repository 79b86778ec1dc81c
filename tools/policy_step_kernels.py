"""One optimiser step of a conv Q-network (forward, loss, backward, fused Adam) at the CLI's batch of 32, run eagerly: after a
warm-up (the library's solver search) the last N steps are traced with torch.profiler and listed kernel by kernel, per step.
  python tools/policy_step_kernels.py [--model UNet|ConvNet] [--steps 20]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import torch
from robotoddler.training.successor_dqn import build_parser, make_nets
ap = argparse.ArgumentParser()
ap.add_argument("--model", default="UNet")
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--plain", action="store_true", help="just run the steps (for an outer profiler: rocprofv3 --pmc)")
a = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)
pol, _ = make_nets(vars(build_parser().parse_args(["--model", a.model])), dev)
pol.train()
opt = torch.optim.Adam(pol.parameters(), lr=1e-4, fused=True)
B = 32
img = lambda p: (torch.rand(B, 1, 64, 64, device=dev) > p).float()
x = (img(0.9), torch.zeros(B, 6, device=dev), img(0.95), torch.rand(1, 1, 64, 64, device=dev).expand(B, -1, -1, -1), img(0.9))
q_t, sf_t = torch.rand(B, device=dev), torch.rand(B, 4096, device=dev)
def step():
    q, sf, _ = pol(*x)
    loss = ((q - q_t) ** 2).mean()
    if sf is not None and a.model == "UNet":
        loss = loss + ((sf[:, 0].reshape(B, -1) - sf_t) ** 2).mean(dim=1).mean()
    opt.zero_grad(set_to_none=True)
    loss.backward()
    opt.step()
    return loss


for _ in range(12):
    step()
torch.cuda.synchronize()
if a.plain:
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
    print(f"{a.model}: {a.steps} steps, last loss {float(loss.detach()):.5f}")
    sys.exit(0)
from torch.profiler import ProfilerActivity, profile
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(a.steps):
        loss = step()
    torch.cuda.synchronize()
rows = sorted(((e.key, e.count, e.device_time_total) for e in prof.key_averages() if e.device_time_total > 0), key=lambda r: -r[2])
total = sum(r[2] for r in rows)
print(f"{a.model}: GPU time per optimiser step {total / a.steps:.1f} us over {sum(r[1] for r in rows) / a.steps:.0f} launches (last loss {float(loss.detach()):.5f})")
for key, count, t in rows:
    print(f"{key[:110]:110s} {count / a.steps:6.1f}/step {t / a.steps:8.1f} us/step {t / count:7.1f} us each")
