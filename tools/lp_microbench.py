"""Micro-benchmark of the stand-alone stability operator on assemblies captured from a random rollout."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "bridges-with-reinforcement-learning_amd")]
import numpy as np, torch
from bridges_hip import abi
from bridges_hip.shapes import load_urdf
from bridges_hip.vec_env import VecAssemblyGym, _ptr, _stream

E = 4096
H = 0.8
env = VecAssemblyGym(E, [load_urdf("shapes/trapezoid.urdf")], [(0.5, 0, i * H + H / 2) for i in range(4)],
                     [(0.5, 0, 4 * H + H / 2)], max_steps=15, seed=0)
for _ in range(40):
    env.select_random(); env.step()
torch.cuda.synchronize()
L = abi.lib()
K = env.K
nb = env.n_blocks.clone()
stable = torch.zeros(E, dtype=torch.uint8, device="cuda")
info = torch.zeros((E, 8), dtype=torch.float64, device="cuda")
ws_stride = abi.lp_ws_stride(K)
ws = torch.zeros((E, ws_stride), dtype=torch.float64, device="cuda")
fixed = torch.zeros(E, dtype=torch.int32, device="cuda")

def run(nb_t, label):
    t = []
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        abi.check(L.bridges_stability(env.table.ptr, E, K, _ptr(env.blk_pose), _ptr(env.blk_verts), _ptr(env.blk_shape),
                                      _ptr(nb_t), _ptr(fixed), 0.8, 1.0, 5.0, 10.0, _ptr(stable), _ptr(info), _ptr(ws),
                                      ws_stride, _stream()))
        torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    inf = info.cpu().numpy(); nbc = nb_t.cpu().numpy()
    print(f"{label}: {min(t)*1e6:.0f} us; pivots mean {inf[:,2].mean():.1f} max {inf[:,2].max():.0f}; n_if max {inf[:,1].max():.0f}; "
          f"nb hist {np.bincount(nbc, minlength=16).tolist()}; errors {int((inf[:,3]!=0).sum())}")
    return inf

prof = hasattr(L, "bridges_debug_lp_profile") and "--profile" in sys.argv
def read_prof(label):
    if not prof:
        return
    buf = (C.c_ulonglong * 8)()
    L.bridges_debug_lp_profile(buf, 1)
    v = list(buf)
    n = max(v[5], 1)
    print(f"   [{label}] cycles per pivot: price {v[0]/n:.0f}  ratio {v[1]/n:.0f}  stage {v[2]/n:.0f}  sweep {v[3]/n:.0f}  (pivots {v[5]})")
read_prof("warm-up")
inf = run(nb, "all envs")
read_prof("all")
for cap in (1, 2, 3, 4, 6, 8):
    run(torch.clamp(nb, max=cap), f"n_blocks clamped to {cap}")
    read_prof(f"clamp {cap}")
for lo, hi in ((0, 2), (2, 4), (4, 6), (6, 9), (9, 16)):
    sel = (nb.cpu().numpy() >= lo) & (nb.cpu().numpy() < hi)
    if sel.any():
        print(f"nb in [{lo},{hi}): n={sel.sum()} cycles interfaces mean {inf[sel,4].mean():.0f} max {inf[sel,4].max():.0f}; "
              f"LP mean {inf[sel,5].mean():.0f} max {inf[sel,5].max():.0f}; pivots mean {inf[sel,2].mean():.1f}; cycles/pivot {inf[sel,5].sum()/max(inf[sel,2].sum(),1):.0f}")
big = inf[:, 2].argsort()[-5:]
print("top pivots", inf[big, 2], "n_if", inf[big, 1], "nb", nb.cpu().numpy()[big])
