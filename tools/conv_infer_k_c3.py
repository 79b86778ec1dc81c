"""ConvNet / Policy inference forward (no_grad) on 2048 rows of 4x64x64: the 32-128-channel layers through the library (default)
against through the training kernel's forward k_c3 (robotoddler.models.cv.K_C3_INFERENCE)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch
from robotoddler.models import cv
from robotoddler.utils.utils import init_weights
dev = torch.device("cuda")
torch.manual_seed(0)
for n in (2048, 512):
    x = [(torch.rand(n, 1, 64, 64, device=dev) > 0.9).float(), torch.zeros(n, 6, device=dev), (torch.rand(n, 1, 64, 64, device=dev) > 0.95).float(),
         torch.rand(n, 1, 64, 64, device=dev), (torch.rand(n, 1, 64, 64, device=dev) > 0.9).float()]
    for name, mk in (("ConvNet", lambda: cv.ConvNet(img_size=(64, 64))), ("Policy", cv.Policy)):
        net = mk().to(dev); net.apply(init_weights); net.eval()
        ref = None
        for flag in (False, True):
            cv.K_C3_INFERENCE = flag
            with torch.no_grad():
                for _ in range(3):
                    out = net(*x)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                for _ in range(8):
                    out = net(*x)
                torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 8
            q = out[0]
            ref = q if ref is None else ref
            print(f"{name} rows={n} k_c3_inference={flag}: {dt*1e3:.2f} ms; max |dq| vs library = {(q-ref).abs().max().item():.2e} (max |q| {ref.abs().max().item():.2e})", flush=True)
