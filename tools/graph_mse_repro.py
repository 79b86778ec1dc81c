"""Tries to reproduce the anomaly recorded in DESIGN.md with torch alone: a multi-workgroup `mean` (F.mse_loss over
131 072 elements) inside a replayed HIP graph returning garbage.  Variants: the loss alone; the loss behind a chain of
ops whose temporaries churn the graph's private memory pool; several losses per graph.  Prints how many of the replays
disagree with the eager value.   python tools/graph_mse_repro.py [--replays 20000]"""
import argparse

import torch
import torch.nn.functional as F

ap = argparse.ArgumentParser()
ap.add_argument("--replays", type=int, default=20000)
ap.add_argument("--sync", action="store_true", help="device synchronisation after every replay")
ap.add_argument("--only", default="", help="substring of the variant names to run")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
B, D = 32, 4096
w = torch.randn(256, 2 * D + 12, device=dev) * 0.05
h = torch.randn(B, 256, device=dev)
t = torch.rand(B, D, device=dev)
out = torch.zeros(64, device=dev)


def variant_plain(k):
    y = h @ w
    out[k] = F.mse_loss(y[:, :D], t)


def variant_churn(k):
    y = h @ w
    for _ in range(4):                                   # temporaries of different sizes come and go
        z = torch.relu(y) * 0.5
        y = y + z[:, : y.shape[1]] * 1e-3
        q = (z[:, :D].softmax(dim=1) * t).sum(dim=1)
    out[k] = F.mse_loss(y[:, :D], t) + 0.0 * q.sum()


def variant_three(k):
    y = h @ w
    out[k] = F.mse_loss(y[:, :D], t)
    out[k + 1] = F.mse_loss(y[:, D:2 * D], t)
    out[k + 2] = ((y[:, :D] - t) ** 2).mean()


y_const = (h @ w).clone()
yc = y_const[:, :D].contiguous()
yc2 = y_const[:, D:2 * D].contiguous()


def variant_three_no_gemm(k):                        # the same three reductions on a constant y: is it the GEMM?
    out[k] = F.mse_loss(y_const[:, :D], t)
    out[k + 1] = F.mse_loss(y_const[:, D:2 * D], t)
    out[k + 2] = ((y_const[:, :D] - t) ** 2).mean()


def variant_three_contiguous(k):                     # ... on contiguous inputs: is it the strided read?
    out[k] = F.mse_loss(yc, t)
    out[k + 1] = F.mse_loss(yc2, t)
    out[k + 2] = ((yc - t) ** 2).mean()


def variant_two_same(k):                             # the same reduction twice
    out[k] = F.mse_loss(yc, t)
    out[k + 1] = F.mse_loss(yc, t)


def variant_sum_only(k):                             # plain sums of a constant tensor, three times
    out[k] = yc.sum()
    out[k + 1] = yc2.sum()
    out[k + 2] = yc.sum()


def variant_sum_add(k):                              # results leave through an element-wise kernel instead of a copy
    out[k:k + 3] = 0
    out[k:k + 1] += yc.sum()
    out[k + 1:k + 2] += yc2.sum()
    out[k + 2:k + 3] += yc.sum()


def variant_sum_kept(k):                             # the three results stay alive until the end (no block is reused)
    r0, r1, r2 = yc.sum(), yc2.sum(), yc.sum()
    out[k] = r0
    out[k + 1] = r1
    out[k + 2] = r2


small, small2 = yc[0, :1024].contiguous(), yc2[0, :1024].contiguous()


def variant_small_sums(k):                           # single-workgroup reductions (no semaphore, no scratch buffers)
    out[k] = small.sum()
    out[k + 1] = small2.sum()
    out[k + 2] = small.sum()


def variant_two_different(k):                        # two different multi-workgroup reductions
    out[k] = yc.sum()
    out[k + 1] = yc2.sum()


def variant_rowwise_then_small(k):                   # the work-around the training graph uses: row-wise, then one small sum
    out[k] = ((yc - t) ** 2).mean(dim=1).mean()
    out[k + 1] = ((yc2 - t) ** 2).mean(dim=1).mean()
    out[k + 2] = ((yc - t) ** 2).mean(dim=1).mean()


for name, fn, n_out in (("mse alone", variant_plain, 1), ("mse behind churning temporaries", variant_churn, 1),
                        ("three reductions per graph", variant_three, 3),
                        ("three reductions, constant y", variant_three_no_gemm, 3),
                        ("three reductions, contiguous", variant_three_contiguous, 3),
                        ("the same mse twice", variant_two_same, 2), ("three plain sums", variant_sum_only, 3),
                        ("three sums, out += (kernel)", variant_sum_add, 3), ("three sums kept alive", variant_sum_kept, 3),
                        ("three single-workgroup sums", variant_small_sums, 3),
                        ("two different multi-workgroup sums", variant_two_different, 2),
                        ("row-wise means, then a small mean", variant_rowwise_then_small, 3)):
    if a.only and a.only not in name:
        continue
    out.zero_()
    fn(0)
    torch.cuda.synchronize()
    ref = out[:n_out].clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(0)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn(0)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    lo = torch.full((n_out,), float("inf"), device=dev)
    hi = torch.full((n_out,), float("-inf"), device=dev)
    first_bad = -1
    for i in range(a.replays):
        g.replay()
        if a.sync:
            torch.cuda.synchronize()
            if first_bad < 0 and not bool(torch.isclose(out[:n_out], ref, rtol=1e-4).all()):
                first_bad = i
        v = out[:n_out]
        bad += (~torch.isclose(v, ref, rtol=1e-4)).sum()          # queued behind the replay, no host sync
        lo = torch.minimum(lo, v)
        hi = torch.maximum(hi, v)
    torch.cuda.synchronize()
    if first_bad >= 0:
        print("   first wrong replay:", first_bad)
    print(f"{name:34s}: {a.replays} replays, {int(bad)} wrong values; eager {ref.tolist()}, replay min {lo.tolist()} max {hi.tolist()}",
          flush=True)
