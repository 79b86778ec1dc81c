"""GPU: the hand-written training passes of the ConvBlock stacks (csrc/conv_train_kernels.hip: conv3x3 forward / input
gradient / weight gradient on the f32 matrix cores, the pooling pair) against PyTorch float32 (the library's convolutions and
autograd) -- operator by operator, then a ConvBlock, then whole optimiser steps of ConvNet (robotoddler/models/cv.py:5-73)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda")

SHAPES = [(5, 4, 64, 16), (32, 16, 64, 16), (3, 16, 32, 32), (32, 32, 32, 32), (4, 32, 16, 64), (32, 64, 16, 64), (32, 64, 8, 128),
          (7, 128, 8, 128), (2, 2, 64, 16), (1, 48, 16, 16), (2, 5, 32, 16), (3, 20, 16, 32), (2, 33, 8, 16)]


def rel(a, b):
    a, b = a.detach(), b.detach()
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def rnd(*shape, seed=0, sparse=0.0):
    g = torch.Generator(device=DEV).manual_seed(seed)
    t = torch.randn(*shape, generator=g, device=DEV)
    if sparse:
        t = t * (torch.rand(*shape, generator=g, device=DEV) > sparse)
    return t


@pytest.mark.parametrize("n,c_in,W,c_out", SHAPES)
def test_conv3x3_forward_modes_match_torch(n, c_in, W, c_out):
    from bridges_hip import dqn_ops
    x, w, b = rnd(n, c_in, W, W, seed=1, sparse=0.5), rnd(c_out, c_in, 3, 3, seed=2) * 0.2, rnd(c_out, seed=3)
    ref = F.conv2d(x, w, padding=1)
    assert rel(dqn_ops.conv3x3(x, w), ref) < 2e-6
    got = dqn_ops.conv3x3(x, w, bias=b)
    want = F.relu(ref + b[None, :, None, None])
    assert rel(got, want) < 2e-6 and bool(((got == 0) == (want == 0)).float().mean() > 0.9999)
    mask = rnd(n, c_out, W, W, seed=4)
    assert rel(dqn_ops.conv3x3(x, w, mask=mask), ref * (mask > 0)) < 2e-6


@pytest.mark.parametrize("n,c_in,W,c_out", [s for s in SHAPES if s[1] % 16 == 0])
def test_conv3x3_input_gradient_matches_torch(n, c_in, W, c_out):
    """transposed=True: dX of a layer c_in -> c_out from the gradient g at its output."""
    from bridges_hip import dqn_ops
    g, w = rnd(n, c_out, W, W, seed=5, sparse=0.6), rnd(c_out, c_in, 3, 3, seed=6) * 0.2
    want = torch.nn.grad.conv2d_input((n, c_in, W, W), w, g, padding=1)
    assert rel(dqn_ops.conv3x3(g, w, transposed=True), want) < 2e-6
    a = rnd(n, c_in, W, W, seed=7)
    assert rel(dqn_ops.conv3x3(g, w, mask=a, transposed=True), want * (a > 0)) < 2e-6


@pytest.mark.parametrize("n,c_in,W,c_out", SHAPES)
def test_conv3x3_weight_gradient_matches_torch_and_is_deterministic(n, c_in, W, c_out):
    from bridges_hip import dqn_ops
    g, x = rnd(n, c_out, W, W, seed=8, sparse=0.6), rnd(n, c_in, W, W, seed=9, sparse=0.5)
    dw, db = dqn_ops.conv3x3_wgrad(g, x)
    want = torch.nn.grad.conv2d_weight(x, (c_out, c_in, 3, 3), g, padding=1)
    assert dw.shape == want.shape and rel(dw, want) < 5e-6
    assert rel(db, g.sum(dim=(0, 2, 3))) < 5e-6
    dw2, db2 = dqn_ops.conv3x3_wgrad(g, x)
    assert torch.equal(dw, dw2) and torch.equal(db, db2)               # fixed summation order: bit-identical


@pytest.mark.parametrize("n,c,W", [(3, 16, 64), (32, 128, 8), (5, 32, 16)])
def test_maxpool2_pair_matches_torch_including_ties(n, c, W):
    from bridges_hip import dqn_ops
    a = F.relu(rnd(n, c, W, W, seed=10))
    a[:, :, ::4, ::4] = a[:, :, 1::4, 1::4]                             # ties between the first and the last cell of many windows
    a = a.contiguous().requires_grad_(True)
    y = F.max_pool2d(a, 2)
    assert torch.equal(dqn_ops.maxpool2(a.detach()), y.detach())
    dy = rnd(n, c, W // 2, W // 2, seed=11)
    # reference: gradient through max_pool2d, times the ReLU mask of a (a = relu(z): d/dz = [a > 0])
    (ga,) = torch.autograd.grad(y, a, dy)
    want = ga * (a.detach() > 0)
    assert torch.equal(dqn_ops.maxpool2_relu_backward(a.detach(), dy), want)


@pytest.mark.parametrize("n,c_in,W,c_out", [(32, 4, 64, 16), (32, 16, 32, 32), (6, 32, 16, 64), (32, 64, 8, 128), (9, 2, 64, 16)])
def test_conv_block_forward_and_backward_match_autograd(n, c_in, W, c_out):
    from robotoddler.models.cv import ConvBlock
    torch.manual_seed(3)
    blk = ConvBlock(c_in, c_out).to(DEV)
    x = rnd(n, c_in, W, W, seed=12, sparse=0.5)
    need_dx = c_in % 16 == 0
    xa, xb = x.clone().requires_grad_(need_dx), x.clone().requires_grad_(need_dx)
    ya = blk.layers(xa)                                                 # the module's own layers: library kernels + autograd
    yb = blk(xb)                                                        # ConvBlock.forward: the hand-written Function
    assert yb.grad_fn is not None and "ConvBlockFunction" in type(yb.grad_fn).__name__
    assert rel(yb, ya) < 2e-6
    dy = rnd(*ya.shape, seed=13)
    params = list(blk.parameters())
    ga = torch.autograd.grad(ya, params + ([xa] if need_dx else []), dy)
    gb = torch.autograd.grad(yb, params + ([xb] if need_dx else []), dy)
    for u, v in zip(gb, ga):
        assert u.shape == v.shape and rel(u, v) < 1e-5, rel(u, v)


def test_convnet_optimiser_steps_follow_the_library_path():
    """Three Adam steps of ConvNet (cv.py:41-73) at the CLI's batch of 32 on 64x64 images: losses within 1e-5 and parameters
    within what two correct float32 Adam runs agree to, hand-written ConvBlocks against the library's."""
    from bridges_hip import dqn_ops
    from robotoddler.models.cv import ConvNet
    from robotoddler.utils.utils import init_weights
    torch.manual_seed(5)
    nets = [ConvNet(img_size=(64, 64)).to(DEV) for _ in range(2)]
    nets[0].apply(init_weights)
    nets[1].load_state_dict(nets[0].state_dict())
    B = 32
    imgs = [(rnd(B, 1, 64, 64, seed=20 + i) > 0.8).float() for i in range(4)]
    binary = (rnd(B, 6, seed=30) > 0).float()
    target = rnd(B, seed=31)
    losses = [[], []]
    for k, net in enumerate(nets):
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        orig = dqn_ops.conv3x3_supported
        if k == 1:
            dqn_ops.conv3x3_supported = lambda *a: False                # the library path (ConvBlock.forward falls back to self.layers)
        try:
            for _ in range(3):
                q = net(imgs[0], binary, imgs[1], imgs[2], imgs[3])[0]
                loss = F.mse_loss(q, target)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses[k].append(float(loss))
        finally:
            dqn_ops.conv3x3_supported = orig
    np.testing.assert_allclose(losses[0], losses[1], rtol=2e-5)
    for pa, pb in zip(nets[0].parameters(), nets[1].parameters()):
        assert rel(pa.detach(), pb.detach()) < 5e-4


@pytest.mark.parametrize("n,c_in,W,c_out", [(32, 4, 64, 16), (8, 16, 64, 16), (32, 64, 32, 32), (5, 32, 16, 64), (32, 32, 64, 16)])
def test_conv3x3_relu_function_matches_autograd(n, c_in, W, c_out):
    """relu(conv(x)) as one Function (the U-Net's layers): output, weight / bias gradient and input gradient against autograd on
    the library's convolution -- the incoming gradient is read through the ReLU's mask inside the two backward kernels."""
    from bridges_hip import dqn_ops
    torch.manual_seed(4)
    conv = torch.nn.Conv2d(c_in, c_out, 3, padding=1).to(DEV)
    x = rnd(n, c_in, W, W, seed=14, sparse=0.4)
    need_dx = c_in % 16 == 0
    xa, xb = x.clone().requires_grad_(need_dx), x.clone().requires_grad_(need_dx)
    ya, yb = torch.relu(conv(xa)), dqn_ops.conv3x3_relu_train(conv, xb)
    assert "Conv3x3ReLUFunction" in type(yb.grad_fn).__name__ and rel(yb.detach(), ya.detach()) < 2e-6
    dy = rnd(*ya.shape, seed=15)
    ga = torch.autograd.grad(ya, list(conv.parameters()) + ([xa] if need_dx else []), dy)
    gb = torch.autograd.grad(yb, list(conv.parameters()) + ([xb] if need_dx else []), dy)
    for u, v in zip(gb, ga):
        assert u.shape == v.shape and rel(u, v) < 1e-5, rel(u, v)


def test_unet_policy_optimiser_steps_follow_the_library_path():
    """Policy (U-Net successor image + ConvNet stability head, cv.py:257-271) with the combined loss: three Adam steps through
    the hand-written conv layers against the library path -- losses 2e-5, parameters to float32 Adam agreement."""
    from bridges_hip import dqn_ops
    from robotoddler.models.cv import Policy
    from robotoddler.utils.utils import init_weights
    torch.manual_seed(6)
    nets = [Policy().to(DEV) for _ in range(2)]
    nets[0].apply(init_weights)
    nets[1].load_state_dict(nets[0].state_dict())
    B = 32
    imgs = [(rnd(B, 1, 64, 64, seed=40 + i) > 0.8).float() for i in range(4)]
    binary = (rnd(B, 6, seed=50) > 0).float()
    q_t, sf_t = rnd(B, seed=51), (rnd(B, 64, 64, seed=52) > 0.5).float()
    losses = [[], []]
    for k, net in enumerate(nets):
        opt = torch.optim.Adam(net.parameters(), lr=1e-4)
        orig = dqn_ops.conv3x3_supported
        if k == 1:
            dqn_ops.conv3x3_supported = lambda *a: False
        try:
            for _ in range(3):
                q, sf, _ = net(imgs[0], binary, imgs[1], imgs[2], imgs[3])
                loss = F.mse_loss(q, q_t) + F.mse_loss(sf[:, 0], sf_t)
                opt.zero_grad()
                loss.backward()
                opt.step()
                losses[k].append(float(loss))
        finally:
            dqn_ops.conv3x3_supported = orig
    np.testing.assert_allclose(losses[0][0], losses[1][0], rtol=1e-5)        # the first loss: forward passes on the same weights
    np.testing.assert_allclose(losses[0], losses[1], rtol=2e-4)              # after one / two Adam updates (losses of O(100))
    for pa, pb in zip(nets[0].parameters(), nets[1].parameters()):
        assert rel(pa.detach(), pb.detach()) < 5e-4


@pytest.mark.parametrize("n,c,hw", [(32, 1, 4096), (32, 16, 4096), (5, 32, 1024), (64, 7, 36)])
def test_bias_grad_is_the_channel_sum_and_deterministic(n, c, hw):
    from bridges_hip import dqn_ops
    x = rnd(n, c, hw, 1, seed=60).requires_grad_(True)
    b = rnd(c, seed=61).requires_grad_(True)
    y = dqn_ops.BiasAddFunction.apply(x, b)
    assert torch.equal(y.detach(), x.detach() + b.detach().view(1, -1, 1, 1))
    dy = rnd(n, c, hw, 1, seed=62)
    dx, db = torch.autograd.grad(y, [x, b], dy)
    exact = dy.double().sum(dim=(0, 2, 3))                  # f32 summation error is relative to the sum of magnitudes
    assert torch.equal(dx, dy)
    assert float(((db.double() - exact).abs() / dy.double().abs().sum(dim=(0, 2, 3))).max()) < 1e-6
    _, db2 = torch.autograd.grad(dqn_ops.BiasAddFunction.apply(x, b), [x, b], dy)
    assert torch.equal(db, db2)


@pytest.mark.parametrize("model", ["ConvNet", "UNet"])
def test_multi_tensor_adam_follows_torch_fused_adam(model):
    """bridges_adam_multi (one launch of 1024-element chunks over every parameter tensor) against torch.optim.Adam(fused=True) on
    the same gradients: parameters, both moments and the step counts after 1 + 3 steps; the torch optimiser carries on from the
    state the launch left (one more torch step on both sides)."""
    from bridges_hip.dqn_ops import MultiTensorAdam
    from robotoddler.training.successor_dqn import build_parser, make_nets
    args = vars(build_parser().parse_args(["--model", model]))
    torch.manual_seed(3)
    a, _ = make_nets(args, torch.device("cuda"))
    b, _ = make_nets(args, torch.device("cuda"))
    b.load_state_dict(a.state_dict())
    oa, ob = (torch.optim.Adam(m.parameters(), lr=1e-3, fused=True) for m in (a, b))
    with pytest.raises(ValueError):
        MultiTensorAdam(ob)                                                   # no state yet: nothing to adopt
    g = torch.Generator(device="cuda").manual_seed(9)
    grads = [[torch.randn(p.shape, device="cuda", generator=g) * (0.1 if k % 2 else 1e-4) for p in a.parameters()] for k in range(5)]

    def set_grads(m, k):
        for p, gr in zip(m.parameters(), grads[k]):
            p.grad = gr.clone()

    for m, o in ((a, oa), (b, ob)):
        set_grads(m, 0)
        o.step()
    mine = MultiTensorAdam(ob)
    assert mine.n_chunks >= sum(p.numel() for p in b.parameters()) // 1024
    for k in (1, 2, 3):
        set_grads(a, k); oa.step()
        set_grads(b, k); mine.step()
    assert float(mine.step_count) == 4.0
    set_grads(a, 4); oa.step()
    set_grads(b, 4); ob.step()                                                # torch takes over from the launch's state
    for pa, pb in zip(a.parameters(), b.parameters()):
        sa, sb = oa.state[pa], ob.state[pb]
        assert float(sa["step"]) == float(sb["step"]) == 5.0
        for x, y in ((pa, pb), (sa["exp_avg"], sb["exp_avg"]), (sa["exp_avg_sq"], sb["exp_avg_sq"])):
            assert torch.allclose(x, y, rtol=2e-5, atol=1e-7), float((x - y).abs().max())


@pytest.mark.parametrize("n,c_in,c_out,H", [(32, 64, 32, 16), (32, 32, 16, 32), (3, 64, 32, 16), (5, 32, 16, 32), (1, 32, 16, 16)])
def test_upconv2x2_training_pair_follows_the_library(n, c_in, c_out, H):
    from bridges_hip import dqn_ops
    up = torch.nn.ConvTranspose2d(c_in, c_out, kernel_size=2, stride=2).cuda()
    x = rnd(n, c_in, H, H, seed=70).requires_grad_(True)
    assert dqn_ops.upconv2x2_train_applies(x, up)
    dy = rnd(n, c_out, 2 * H, 2 * H, seed=71)
    y = dqn_ops.conv_bias_train(up, x)
    ref = up(x)
    assert rel(y, ref) < 3e-6
    got = torch.autograd.grad(y, [x, up.weight, up.bias], dy)
    want = torch.autograd.grad(ref, [x, up.weight, up.bias], dy)
    for g_, w_, name in zip(got, want, ("dx", "dw", "db")):
        assert g_.shape == w_.shape and rel(g_, w_) < 2e-5, (name, rel(g_, w_))
    again = torch.autograd.grad(dqn_ops.conv_bias_train(up, x), [x, up.weight, up.bias], dy)
    assert all(torch.equal(a, b) for a, b in zip(got, again))                  # deterministic


@pytest.mark.parametrize("n,c_in,H", [(32, 16, 64), (3, 16, 64), (2, 7, 10), (300, 16, 64)])
def test_conv1x1_to_one_channel_follows_the_library(n, c_in, H):
    from bridges_hip import dqn_ops
    conv = torch.nn.Conv2d(c_in, 1, kernel_size=1).cuda()
    x = rnd(n, c_in, H, H, seed=72).requires_grad_(True)
    assert dqn_ops.conv1x1_o1_applies(x, conv)
    dy = rnd(n, 1, H, H, seed=73)
    y = dqn_ops.conv_bias_train(conv, x)
    ref = conv(x)
    assert y.shape == ref.shape and rel(y, ref) < 3e-6
    got = torch.autograd.grad(y, [x, conv.weight, conv.bias], dy)
    want = torch.autograd.grad(ref, [x, conv.weight, conv.bias], dy)
    for g_, w_, name in zip(got, want, ("dx", "dw", "db")):
        assert g_.shape == w_.shape and rel(g_, w_) < 2e-5, (name, rel(g_, w_))
    again = torch.autograd.grad(dqn_ops.conv_bias_train(conv, x), [x, conv.weight, conv.bias], dy)
    assert all(torch.equal(a, b) for a, b in zip(got, again))


def test_misaligned_image_views_are_cloned_by_the_wrappers_and_refused_by_the_c_abi():
    """The conv kernels read image rows as float4: a view that starts 4 bytes into an allocation goes through a clone in
    dqn_ops (same result), and the C entry points refuse such a pointer instead of faulting."""
    import ctypes as C
    from bridges_hip import abi, dqn_ops
    n, c, W, co = 3, 16, 32, 32
    flat = rnd(n * c * W * W + 1, seed=80)
    x_mis = flat[1:].view(n, c, W, W)
    assert x_mis.data_ptr() % 16 != 0 and x_mis.is_contiguous()
    x = x_mis.clone()
    w, b = rnd(co, c, 3, 3, seed=81) * 0.2, rnd(co, seed=82)
    assert torch.equal(dqn_ops.conv3x3(x_mis, w, b), dqn_ops.conv3x3(x, w, b))
    g = rnd(n, co, W, W, seed=83)
    for a_, b_ in zip(dqn_ops.conv3x3_wgrad(g, x_mis), dqn_ops.conv3x3_wgrad(g, x)):
        assert torch.equal(a_, b_)
    out = torch.empty(n, co, W, W, device=DEV)
    L = abi.lib()
    p = lambda t_: C.c_void_p(t_.data_ptr())
    rc = L.bridges_conv3x3(p(x_mis), None, p(w), p(b), None, p(out), n, c, co, W, 1, 0, None)
    assert rc != 0 and "aligned" in L.bridges_last_error().decode()
    with pytest.raises(abi.BridgesHipError, match="aligned"):
        abi.check(rc, "bridges_conv3x3")


@pytest.mark.parametrize("model", ["ConvNet", "UNet"])
def test_deferred_weight_gradient_reduction_gives_the_same_bits(model):
    """with dqn_ops.deferred_wgrad_reduce(tables): loss.backward() -- the conv3x3 layers' partial sums added by ONE launch at the
    end of the pass instead of one per layer: every parameter's .grad bit-identical to the per-layer form; a second pass through
    the same tables as well; a parameter whose .grad already exists keeps the immediate form (accumulation stays correct)."""
    from bridges_hip import dqn_ops
    from robotoddler.training.successor_dqn import build_parser, make_nets
    args = vars(build_parser().parse_args(["--model", model]))
    torch.manual_seed(4)
    net, _ = make_nets(args, DEV)
    net.train()
    B = 8
    img = lambda p, s: (torch.rand(B, 1, 64, 64, device=DEV, generator=torch.Generator(device=DEV).manual_seed(s)) > p).float()
    x = (img(0.9, 1), torch.zeros(B, 6, device=DEV), img(0.95, 2), torch.rand(1, 1, 64, 64, device=DEV).expand(B, -1, -1, -1), img(0.9, 3))

    def loss():
        q, sf, _ = net(*x)
        out = (q ** 2).mean()
        return out + (sf ** 2).mean() if sf is not None else out

    net.zero_grad(set_to_none=True)
    loss().backward()
    want = [p.grad.clone() for p in net.parameters()]
    tables = dqn_ops.ReduceTables(DEV)
    for _ in range(2):
        net.zero_grad(set_to_none=True)
        with dqn_ops.deferred_wgrad_reduce(tables) as q:
            loss().backward()
        assert len(q.jobs) == (8 if model == "ConvNet" else 21)          # + the U-Net's two transposed and its 1x1 convolution
        for p, w in zip(net.parameters(), want):
            assert torch.equal(p.grad, w)
    with dqn_ops.deferred_wgrad_reduce(tables) as q:                          # .grad exists: nothing is deferred, gradients add up
        loss().backward()
    assert len(q.jobs) == 0
    for p, w in zip(net.parameters(), want):
        assert torch.allclose(p.grad, 2 * w, rtol=1e-6, atol=1e-12)
