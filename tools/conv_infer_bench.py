"""ConvNet / Policy inference forward (no_grad) on 2048 rows of 4x64x64: NCHW vs channels_last, cudnn.benchmark on/off."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import torch
from robotoddler.models.cv import ConvNet, Policy
from robotoddler.utils.utils import init_weights
dev = torch.device("cuda")
torch.manual_seed(0)
n = 2048
x = [(torch.rand(n, 1, 64, 64, device=dev) > 0.9).float(), torch.zeros(n, 6, device=dev), (torch.rand(n, 1, 64, 64, device=dev) > 0.95).float(),
     torch.rand(n, 1, 64, 64, device=dev), (torch.rand(n, 1, 64, 64, device=dev) > 0.9).float()]
for name, mk in (("ConvNet", lambda: ConvNet(img_size=(64, 64))), ("Policy", Policy)):
    net = mk().to(dev); net.apply(init_weights); net.eval()
    ref = None
    for bench in (False, True):
        torch.backends.cudnn.benchmark = bench
        for cl in (False, True):
            m = mk().to(dev); m.load_state_dict(net.state_dict()); m.eval()
            if cl:
                m = m.to(memory_format=torch.channels_last)
            with torch.no_grad():
                for _ in range(3):
                    out = m(*x)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(5):
                    out = m(*x)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            q = out[0]
            if ref is None:
                ref = q
            print(f"{name} benchmark={bench} channels_last={cl}: {dt*1e3:.2f} ms per 2048 rows; max |dq| vs first = {(q-ref).abs().max().item():.2e}", flush=True)
