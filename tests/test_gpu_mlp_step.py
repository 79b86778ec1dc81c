"""GPU: the hand-written SuccessorMLP training step (csrc/mlp_kernels.hip: skinny GEMMs on the f32 matrix cores, head +
loss + gradient, backward) against torch -- the linear layers against float64 products, the whole step against
autograd on the module's own forward (robotoddler/models/cv.py:76-105) and the losses of train_policy_net
(successor_dqn.py:215-232).  Tolerance: 1e-5 relative to the largest entry (BASELINE.json: float features / losses)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(a, b):
    b = b.double()
    return float((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("rows,K,N", [(32, 16390, 256), (32, 256, 8204), (32, 256, 128), (64, 70, 33), (32, 1030, 40),
                                      (32, 64, 128), (96, 523, 257)])
def test_linear_forward_and_backward_match_float64_products(rows, K, N):
    from bridges_hip import mlp_ops
    g = torch.Generator(device=DEV).manual_seed(rows * 7 + K + N)
    x = torch.randn(rows, K, device=DEV, generator=g)
    w = torch.randn(N, K, device=DEV, generator=g) / np.sqrt(K)
    b = torch.randn(N, device=DEV, generator=g)
    ws = torch.empty(1 << 20, device=DEV)
    for relu in (False, True):
        y = mlp_ops.linear_forward(x, w, b, relu, ws=ws)
        ref = x.double() @ w.double().T + b.double()
        ref = ref.clamp_min(0) if relu else ref
        assert rel_err(y, ref) < 2e-6, (relu, rel_err(y, ref))
    # a workspace too small for the preferred split count still gives the same numbers
    y_small = mlp_ops.linear_forward(x, w, b, False, ws=torch.empty(rows * N * 2, device=DEV))
    assert rel_err(y_small, x.double() @ w.double().T + b.double()) < 2e-6
    dz = torch.randn(rows, N, device=DEV, generator=g)
    act = torch.randn(rows, K, device=DEV, generator=g)                  # stands for the ReLU output below
    dW, db, below = mlp_ops.linear_backward(dz, x, w, act_below=act, ws=ws)
    assert rel_err(dW, dz.double().T @ x.double()) < 2e-6
    assert rel_err(db, dz.double().sum(0)) < 2e-6
    ref_below = (dz.double() @ w.double()) * (act > 0)
    assert rel_err(below, ref_below) < 2e-6
    assert bool((below[act <= 0] == 0).all())
    dW2, db2, none = mlp_ops.linear_backward(dz, x, w, need_input_grad=False, ws=ws)
    assert none is None and torch.equal(dW2, dW) and torch.equal(db2, db)      # deterministic


def make_net(hidden=(256, 128, 64, 128, 256), size=64, seed=0):
    from robotoddler.models.cv import SuccessorMLP
    from robotoddler.utils.utils import init_weights
    torch.manual_seed(seed)
    net = SuccessorMLP(img_size=(size, size), hidden_dims=list(hidden)).to(DEV)
    net.apply(init_weights)
    return net


def make_batch(n, size, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    px = size * size
    block = (torch.rand(n, 1, size, size, device=DEV, generator=g) < 0.05).float()
    action = (torch.rand(n, 1, size, size, device=DEV, generator=g) < 0.01).float()
    binary = (torch.rand(n, 6, device=DEV, generator=g) < 0.5).float()
    reward = torch.rand(1, size, size, device=DEV, generator=g)
    obstacle = (torch.rand(1, size, size, device=DEV, generator=g) < 0.03).float()
    q_t = torch.randn(n, device=DEV, generator=g) * 3
    sf_t = torch.rand(n, px, device=DEV, generator=g)
    return block, action, binary, reward, obstacle, q_t, sf_t


def autograd_step(net, batch, rows, use_q, use_sf):
    block, action, binary, reward, obstacle, q_t, sf_t = batch
    B = rows.stop - rows.start
    for p in net.parameters():
        p.grad = None
    q, sf, _ = net(block[rows], binary[rows], action[rows], reward.unsqueeze(0).expand(B, -1, -1, -1),
                   obstacle.unsqueeze(0).expand(B, -1, -1, -1))
    mse = torch.nn.MSELoss()
    loss = 0.
    if use_q:
        loss = loss + mse(q, q_t[rows])
    if use_sf:
        loss = loss + mse(sf[:, 0].reshape(B, -1), sf_t[rows])
    loss.backward()
    return float(loss.detach()), q.detach(), [p.grad.clone() for p in net.parameters()]


@pytest.mark.parametrize("B,size,hidden,use_q,use_sf", [(32, 64, (256, 128, 64, 128, 256), True, True),
                                                        (32, 64, (256, 128, 64, 128, 256), False, True),
                                                        (16, 64, (256, 128, 64, 128, 256), True, False),
                                                        (4, 64, (128, 64, 128), True, True),
                                                        (48, 32, (96, 40), True, True)])
def test_fused_step_matches_autograd(B, size, hidden, use_q, use_sf):
    from bridges_hip.mlp_ops import FusedSuccessorStep
    net = make_net(hidden, size, seed=B)
    n_batches = 3
    batch = make_batch(n_batches * B, size, seed=B + size)
    block, action, binary, reward, obstacle, q_t, sf_t = batch
    refs = [autograd_step(net, batch, slice(i * B, (i + 1) * B), use_q, use_sf) for i in range(n_batches)]
    for p in net.parameters():
        p.grad = None
    fused = FusedSuccessorStep(net, B, use_q, use_sf)
    counter = torch.zeros((), dtype=torch.int64, device=DEV)
    losses = torch.zeros(n_batches, device=DEV)
    px = size * size
    for i in range(n_batches):
        fused.launch(counter, block.reshape(-1, px), action.reshape(-1, px), binary, reward.reshape(px).contiguous(),
                     obstacle.reshape(px).contiguous(), q_t, sf_t, losses)
        loss_ref, q_ref, grads_ref = refs[i]
        assert int(counter) == i + 1
        assert abs(float(losses[i]) - loss_ref) <= 1e-5 * max(1.0, abs(loss_ref)), (i, float(losses[i]), loss_ref)
        assert rel_err(fused.q[:B], q_ref) < 1e-5
        for p, gref in zip(net.parameters(), grads_ref):
            assert rel_err(p.grad, gref) < 1e-5, (i, tuple(p.shape), rel_err(p.grad, gref))


def test_three_adam_steps_follow_the_autograd_run():
    """Three optimiser steps (fused Adam) fed by the hand-written backward against three fed by autograd."""
    from bridges_hip.mlp_ops import FusedSuccessorStep
    B, size = 32, 64
    batch = make_batch(3 * B, size, seed=11)
    block, action, binary, reward, obstacle, q_t, sf_t = batch
    px = size * size
    net_a, net_b = make_net(seed=3), make_net(seed=3)
    opt_a = torch.optim.Adam(net_a.parameters(), lr=1e-3, fused=True)
    opt_b = torch.optim.Adam(net_b.parameters(), lr=1e-3, fused=True)
    losses_a = []
    for i in range(3):
        loss, _, grads = autograd_step(net_a, batch, slice(i * B, (i + 1) * B), True, True)
        opt_a.step()
        losses_a.append(loss)
    fused = FusedSuccessorStep(net_b, B, True, True)
    counter = torch.zeros((), dtype=torch.int64, device=DEV)
    losses_b = torch.zeros(3, device=DEV)
    for i in range(3):
        fused.launch(counter, block.reshape(-1, px), action.reshape(-1, px), binary, reward.reshape(px).contiguous(),
                     obstacle.reshape(px).contiguous(), q_t, sf_t, losses_b)
        opt_b.step()
    np.testing.assert_allclose(losses_b.cpu().numpy(), np.array(losses_a), rtol=2e-5)
    # Adam divides by sqrt(v): where a gradient entry is rounding noise its first updates are +-lr whatever its size, so
    # the parameters of two correct float32 runs agree to ~lr * 1e-2 only, not to 1e-5 (the gradients themselves do)
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert rel_err(pb.detach(), pa.detach()) < 2e-4


def test_flat_adam_launch_follows_torch_adam():
    """bridges_adam_step as the last launch of the sequence (FusedSuccessorStep(optimizer=...)): three optimiser steps
    against torch.optim.Adam fed by autograd on an identically initialised net; the optimiser's own state tensors are the
    flat moment buffers and export_state() hands the step count back."""
    from bridges_hip.dqn_ops import FlatParameters
    from bridges_hip.mlp_ops import FusedSuccessorStep
    B, size = 32, 64
    batch = make_batch(3 * B, size, seed=5)
    block, action, binary, reward, obstacle, q_t, sf_t = batch
    px = size * size
    net_a, net_b = make_net(seed=9), make_net(seed=9)
    net_b._flat_params = FlatParameters(net_b)
    opt_a = torch.optim.Adam(net_a.parameters(), lr=1e-3, fused=True)
    opt_b = torch.optim.Adam(net_b.parameters(), lr=1e-3, fused=True)
    for i in range(3):
        autograd_step(net_a, batch, slice(i * B, (i + 1) * B), True, True)
        opt_a.step()
    fused = FusedSuccessorStep(net_b, B, True, True, optimizer=opt_b)
    assert fused.fused_adam
    counter = torch.zeros((), dtype=torch.int64, device=DEV)
    losses_b = torch.zeros(3, device=DEV)
    for i in range(3):
        fused.launch(counter, block.reshape(-1, px), action.reshape(-1, px), binary, reward.reshape(px).contiguous(),
                     obstacle.reshape(px).contiguous(), q_t, sf_t, losses_b)
    assert int(counter) == 3 and float(fused.adam_step) == 3.0 and int(fused.ticket[0]) == 0
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert rel_err(pb.detach(), pa.detach()) < 2e-4
        sa, sb = opt_a.state[pa], opt_b.state[pb]
        assert rel_err(sb["exp_avg"], sa["exp_avg"]) < 1e-4 and rel_err(sb["exp_avg_sq"], sa["exp_avg_sq"]) < 1e-4
    fused.export_state()
    assert all(float(st["step"]) == 3.0 for st in opt_b.state.values())
    # a fourth step by torch's own optimizer.step() continues from the exported state (same moments, step 4)
    autograd_step(net_a, batch, slice(0, B), True, True)
    opt_a.step()
    for p in net_b.parameters():
        p.grad = None
    autograd_step(net_b, batch, slice(0, B), True, True)
    opt_b.step()
    for pa, pb in zip(net_a.parameters(), net_b.parameters()):
        assert rel_err(pb.detach(), pa.detach()) < 3e-4
    # bridges_adam_step alone against torch's formula on random data (one step, t = 7)
    from bridges_hip import abi
    from bridges_hip.ops import _ptr, _stream
    g = torch.Generator(device=DEV).manual_seed(1)
    n = 4099
    p0 = torch.randn(n + 1, device=DEV, generator=g)[:n].contiguous()
    gr, m0 = torch.randn(n, device=DEV, generator=g), torch.randn(n, device=DEV, generator=g) * 0.1
    v0 = torch.rand(n, device=DEV, generator=g) * 0.01
    p1, m1, v1 = p0.clone(), m0.clone(), v0.clone()
    step = torch.full((), 7.0, device=DEV)
    abi.check(abi.lib().bridges_adam_step(_ptr(p1), _ptr(gr), _ptr(m1), _ptr(v1), n, _ptr(step), 1e-3, 0.9, 0.999, 1e-8,
                                          _stream()), "bridges_adam_step")
    md = m0.double() + (gr.double() - m0.double()) * (1 - 0.9)
    vd = 0.999 * v0.double() + (1 - 0.999) * gr.double() ** 2
    pd = p0.double() - (1e-3 / (1 - 0.9 ** 7)) * md / (vd.sqrt() / (1 - 0.999 ** 7) ** 0.5 + 1e-8)
    assert rel_err(m1, md.float()) < 1e-6 and rel_err(v1, vd.float()) < 1e-6 and rel_err(p1, pd.float()) < 2e-6


def test_prebuilt_input_rows_give_the_same_steps():
    """allocate_inputs / prepare_inputs (all batches' first-layer rows in one launch, a step reads block `counter` of them)
    against the per-step build of the rows: identical losses, q values and gradients, bit for bit."""
    from bridges_hip.mlp_ops import FusedSuccessorStep
    B, size, n_batches = 32, 64, 3
    batch = make_batch(n_batches * B, size, seed=21)
    block, action, binary, reward, obstacle, q_t, sf_t = batch
    px = size * size
    rw, ob = reward.reshape(px).contiguous(), obstacle.reshape(px).contiguous()
    out = {}
    for mode in ("per_step", "prebuilt"):
        net = make_net(seed=4)
        fused = FusedSuccessorStep(net, B, True, True)
        if mode == "prebuilt":
            fused.allocate_inputs(n_batches)
            fused.prepare_inputs(n_batches, block.reshape(-1, px), action.reshape(-1, px), binary, rw, ob)
        counter = torch.zeros((), dtype=torch.int64, device=DEV)
        losses = torch.zeros(n_batches, device=DEV)
        rec = []
        for i in range(n_batches):
            fused.launch(counter, block.reshape(-1, px), action.reshape(-1, px), binary, rw, ob, q_t, sf_t, losses)
            rec.append((fused.q[:B].clone(), [p.grad.clone() for p in net.parameters()]))
        out[mode] = (losses.clone(), rec)
    assert torch.equal(out["per_step"][0], out["prebuilt"][0])
    for (qa, ga), (qb, gb) in zip(out["per_step"][1], out["prebuilt"][1]):
        assert torch.equal(qa, qb)
        for x, y in zip(ga, gb):
            assert torch.equal(x, y)


@pytest.mark.parametrize("hidden,with_adam", [((256, 128, 64, 128, 256), False), ((256, 128, 64, 128, 256), True),
                                              ((128, 64, 32, 256), False)])
def test_middle_layer_stack_equals_the_per_layer_launches(hidden, with_adam, monkeypatch):
    """bridges_mlp_mid_forward / _backward (the middle Linear + ReLU layers as one launch each way, a workgroup per tile of the
    stack's last layer) against a launch per layer (mid_stack=False): losses, q values, every gradient and -- with the Adam
    update in the step -- every weight bit for bit over three steps.  A stack the library is not built for keeps the per-layer
    launches."""
    from bridges_hip.mlp_ops import FusedSuccessorStep
    B, size, n_batches = 32, 64, 3
    batch = make_batch(n_batches * B, size, seed=33)
    block, action, binary, reward, obstacle, q_t, sf_t = batch
    px = size * size
    rw, ob = reward.reshape(px).contiguous(), obstacle.reshape(px).contiguous()
    out = {}
    for mode in ("per_layer", "stack"):
        net = make_net(hidden=hidden, seed=6)
        opt = None
        if with_adam:                                           # the update inside the last launch: weights change every step
            from bridges_hip.dqn_ops import FlatParameters
            net._flat_params = FlatParameters(net)
            opt = torch.optim.Adam(net.parameters(), lr=1e-3, fused=True)
        fused = FusedSuccessorStep(net, B, True, True, optimizer=opt, mid_stack=(mode == "stack"))
        assert (fused.mid is not None) == (mode == "stack" and hidden == (256, 128, 64, 128, 256)) and fused.fused_adam == with_adam
        counter = torch.zeros((), dtype=torch.int64, device=DEV)
        losses = torch.zeros(n_batches, device=DEV)
        rec = []
        for i in range(n_batches):
            fused.launch(counter, block.reshape(-1, px), action.reshape(-1, px), binary, rw, ob, q_t, sf_t, losses)
            rec.append((fused.q[:B].clone(), [p.grad.clone() for p in net.parameters()] + [p.detach().clone() for p in net.parameters()]))
        out[mode] = (losses.clone(), rec)
    assert torch.equal(out["per_layer"][0], out["stack"][0])
    for (qa, ga), (qb, gb) in zip(out["per_layer"][1], out["stack"][1]):
        assert torch.equal(qa, qb)
        for x, y in zip(ga, gb):
            assert torch.equal(x, y)


@pytest.mark.parametrize("n", [1, 32, 45, 1000, 8209])
def test_middle_stack_for_many_rows_equals_the_layer_by_layer_forward(n):
    """bridges_mlp_mid_rows (the acting forward's middle layers, two launches) against bridges_linear_forward layer by layer on
    the same rows: bit for bit (same tiles, same order), and within 1e-5 of the float64 product; rows beyond n untouched."""
    from bridges_hip import mlp_ops
    net = make_net(seed=11)
    lin = [m for m in net.mlp.layers if isinstance(m, torch.nn.Linear)]
    g = torch.Generator(device="cpu").manual_seed(n)
    h_pre = torch.randn((n, 256), generator=g).to(DEV)
    got = mlp_ops.mid_rows(h_pre, lin[1:-1])
    assert got is not None and tuple(got.shape) == (n, 256)
    rows = 32 * ((n + 31) // 32)
    a = torch.zeros((rows, 256), device=DEV)
    a[:n] = torch.relu(h_pre)
    ref64 = torch.relu(h_pre).double()
    for l in lin[1:-1]:
        a = mlp_ops.linear_forward(a, l.weight, l.bias, relu=True)
        ref64 = torch.relu(ref64 @ l.weight.double().T + l.bias.double())
    assert torch.equal(got, a[:n])
    assert rel_err(got, ref64.float()) < 1e-5
    # a strided view as input (rows of a wider buffer)
    wide = torch.zeros((n, 320), device=DEV)
    wide[:, :256] = h_pre
    assert torch.equal(mlp_ops.mid_rows(wide[:, :256], lin[1:-1]), got)
