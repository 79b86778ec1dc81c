"""Policy (U-Net + ConvNet) inference forward on 2048 rows, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import torch
from robotoddler.models.cv import Policy
from robotoddler.utils.utils import init_weights
dev = torch.device("cuda")
torch.manual_seed(0)
n = 2048
x = [(torch.rand(n, 1, 64, 64, device=dev) > 0.9).float(), torch.zeros(n, 6, device=dev), (torch.rand(n, 1, 64, 64, device=dev) > 0.95).float(),
     torch.rand(n, 1, 64, 64, device=dev), (torch.rand(n, 1, 64, 64, device=dev) > 0.9).float()]
net = Policy().to(dev); net.apply(init_weights); net.eval()
import time
with torch.no_grad():
    for _ in range(4):
        out = net(*x)                 # MIOpen's solver search happens here
    torch.cuda.synchronize()
    print("MARK_BEGIN_NS", time.time_ns(), flush=True)
    for _ in range(8):
        out = net(*x)
    torch.cuda.synchronize()
print("done")
