"""Host wrappers of the fused DQN ops (K7): TD / successor-feature target construction and Polyak soft update."""
import ctypes as C

import numpy as np

import torch

from . import abi


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


_stream = abi.current_stream


def td_target(seg_offset, next_q, lin_reward, done, gamma, next_sf=None, action_raster=None):
    """Target construction of train_policy_net (successor_dqn.py:197-213, 222, 230) in one kernel.

    seg_offset int32 [B+1]: rows [seg_offset[i], seg_offset[i+1]) of next_q / next_sf belong to transition i; or a pair
    (lo, hi) of int32 [B] tensors with the rows [lo[i], hi[i]) (transitions whose next states are the same state share rows).
    next_q f32 [R]; next_sf f32 [R, D] view (row stride may exceed D, e.g. psi[:, 0] of a [R,2,64,64] tensor);
    action_raster f32 [B, D]; lin_reward f32 [B]; done uint8/bool [B].
    Returns (q_target [B], sf_target [B, D] or None, argmax_row int32 [B])."""
    L = abi.require_gpu()
    dev = next_q.device
    assert next_q.is_contiguous() and next_q.dtype == torch.float32
    if isinstance(seg_offset, (tuple, list)):
        seg_lo, seg_hi = (t.to(device=dev, dtype=torch.int32).contiguous() for t in seg_offset)
        B = seg_lo.numel()
        assert seg_hi.numel() == B
    else:
        seg_offset = seg_offset.to(device=dev, dtype=torch.int32).contiguous()
        B = seg_offset.numel() - 1
        seg_lo, seg_hi = seg_offset[:B], seg_offset[1:]                # views of one buffer: (seg, seg + 1)
    lin_reward = lin_reward.to(device=dev, dtype=torch.float32).contiguous().reshape(-1)
    done_u8 = done.to(device=dev).to(torch.uint8).contiguous()
    q_target = torch.empty(B, dtype=torch.float32, device=dev)
    argmax_row = torch.empty(B, dtype=torch.int32, device=dev)
    sf_dim, sf_target, stride = 0, None, 0
    if next_sf is not None:
        D = next_sf[0].numel()
        stride = next_sf.stride(0)
        assert next_sf.dtype == torch.float32 and next_sf[0].is_contiguous()
        action_raster = action_raster.to(torch.float32).reshape(B, D).contiguous()
        sf_target = torch.empty((B, D), dtype=torch.float32, device=dev)
        sf_dim = D
    abi.check(L.bridges_td_target(B, _ptr(seg_lo), _ptr(seg_hi), _ptr(next_q), _ptr(next_sf), stride, _ptr(action_raster),
                                  _ptr(lin_reward), _ptr(done_u8), float(gamma), sf_dim, _ptr(q_target), _ptr(sf_target),
                                  _ptr(argmax_row), _stream()), "bridges_td_target")
    return q_target, sf_target, argmax_row


def soft_update_(target, policy, tau):
    """target <- policy * tau + target * (1 - tau) in place (successor_dqn.py:280-288); float32 contiguous tensors."""
    L = abi.require_gpu()
    assert target.dtype == torch.float32 and policy.dtype == torch.float32
    assert target.is_contiguous() and policy.is_contiguous() and target.numel() == policy.numel()
    abi.check(L.bridges_soft_update(_ptr(target), _ptr(policy), target.numel(), float(tau), float(1.0 - tau), _stream()),
              "bridges_soft_update")
    return target


def bias_relu_(x, bias):
    """relu(x + bias[c]) in place on the NCHW output ``x`` [n, C, H, W] of a bias-free convolution (one pass)."""
    L = abi.require_gpu()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and bias.dtype == torch.float32
    n, C, H, W = x.shape
    abi.check(L.bridges_bias_relu(_ptr(x), _ptr(bias.contiguous()), n, C, H * W, _stream()), "bridges_bias_relu")
    return x


def bias_relu_pool2(x, bias):
    """maxpool2(relu(x + bias[c])) of the NCHW output of a bias-free convolution in one pass -> [n, C, H/2, W/2]."""
    L = abi.require_gpu()
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and bias.dtype == torch.float32
    n, C, H, W = x.shape
    out = torch.empty((n, C, H // 2, W // 2), dtype=torch.float32, device=x.device)
    abi.check(L.bridges_bias_relu_pool2(_ptr(x), _ptr(bias.contiguous()), _ptr(out), n, C, H, W, _stream()),
              "bridges_bias_relu_pool2")
    return out


def conv3x3_relu_o16_applies(x, conv, x2=None):
    """The hand-written conv3x3 + bias + ReLU [+ pool] kernel covers: 16 output channels, 1-4 / 16 / 32 input channels,
    64-pixel-wide float32 NCHW images with a height that is a multiple of 8, stride 1, padding 1 (x2: the input is the
    concatenation of two 16-channel tensors)."""
    if x2 is not None and not (x.shape[1] == 16 and x2.shape[1] == 16 and conv.in_channels == 32 and x2.shape[2:] == x.shape[2:]):
        return False
    return (conv.out_channels == 16 and conv.in_channels in (1, 2, 3, 4, 16, 32) and tuple(conv.kernel_size) == (3, 3)
            and tuple(conv.padding) == (1, 1) and tuple(conv.stride) == (1, 1) and tuple(conv.dilation) == (1, 1)
            and conv.groups == 1 and conv.bias is not None and x.dim() == 4 and x.shape[3] == 64 and x.shape[2] % 8 == 0
            and x.dtype == torch.float32 and x.is_cuda)


def conv3x3_relu_o16(x, weight, bias, pool=False, x2=None, both=False, proj=None):
    """relu(conv2d(x, weight, bias, padding=1)) by bridges_conv3x3_relu_o16_ex (f32 matrix cores), with the neighbours
    the U-Net puts around it folded in:
      pool=True        -> max_pool2d(., 2) of it;
      both=True        -> (it, max_pool2d(it, 2));
      x2=tensor        -> the input is torch.cat([x, x2], dim=1) (16 + 16 channels), not materialised;
      proj=(w1, b1)    -> conv2d(it, w1, b1) for a 1x1 convolution w1 [1,16,1,1] to one channel."""
    L = abi.require_gpu()
    x = x.contiguous()
    n, c_in, H, W = x.shape
    c_in2 = 0
    if x2 is not None:
        x2 = x2.contiguous()
        assert x2.shape[0] == n and x2.shape[2:] == x.shape[2:]
        c_in2 = x2.shape[1]
    dev = x.device
    mode, out2, pw, pb = 0, None, None, None
    if proj is not None:
        mode, pw, pb = 3, proj[0].reshape(-1).contiguous(), proj[1].reshape(-1).contiguous()
        assert pw.numel() == 16 and pb.numel() == 1 and not (pool or both)
        out = torch.empty((n, 1, H, W), dtype=torch.float32, device=dev)
    elif both:
        mode = 2
        out = torch.empty((n, 16, H, W), dtype=torch.float32, device=dev)
        out2 = torch.empty((n, 16, H // 2, W // 2), dtype=torch.float32, device=dev)
    elif pool:
        mode = 1
        out = torch.empty((n, 16, H // 2, W // 2), dtype=torch.float32, device=dev)
    else:
        out = torch.empty((n, 16, H, W), dtype=torch.float32, device=dev)
    abi.check(L.bridges_conv3x3_relu_o16_ex(_ptr(x), _ptr(x2), _ptr(weight.contiguous()), _ptr(bias.contiguous()), _ptr(out),
                                            _ptr(out2), _ptr(pw), _ptr(pb), n, c_in, c_in2, H, W, mode, _stream()),
              "bridges_conv3x3_relu_o16")
    return (out, out2) if both else out


def upconv2x2_applies(x, up):
    """ConvTranspose2d(kernel 2, stride 2) layers the hand-written kernel covers (the U-Net's upconv3 / upconv4)."""
    return ((up.in_channels, up.out_channels) in ((32, 16), (64, 32)) and tuple(up.kernel_size) == (2, 2)
            and tuple(up.stride) == (2, 2) and tuple(up.padding) == (0, 0) and tuple(up.output_padding) == (0, 0)
            and up.groups == 1 and up.bias is not None and x.dim() == 4 and x.shape[3] % 16 == 0
            and x.dtype == torch.float32 and x.is_cuda)


def upconv2x2(x, weight, bias):
    """conv_transpose2d(x, weight, bias, stride=2) for a 2x2 kernel by bridges_upconv2x2 (f32 matrix cores)."""
    L = abi.require_gpu()
    x = x.contiguous()
    n, c_in, H, W = x.shape
    c_out = weight.shape[1]
    out = torch.empty((n, c_out, 2 * H, 2 * W), dtype=torch.float32, device=x.device)
    abi.check(L.bridges_upconv2x2(_ptr(x), _ptr(weight.contiguous()), _ptr(bias.contiguous()), _ptr(out), n, c_in, c_out, H, W,
                                  _stream()), "bridges_upconv2x2")
    return out


def conv3x3_supported(x, c_out):
    """Shapes bridges_conv3x3 / bridges_conv3x3_wgrad cover: float32 NCHW on the GPU, square images of 8 / 16 / 32 / 64
    pixels, C_out a multiple of 16."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[2] == x.shape[3] and x.shape[3] in (8, 16, 32, 64)
            and c_out % 16 == 0 and c_out >= 16)


def _vec4(t):
    """contiguous and 16-byte aligned (the conv kernels read image rows as float4)."""
    t = t.contiguous()
    return t if t.data_ptr() % 16 == 0 else t.clone()


def conv3x3(x, weight, bias=None, mask=None, transposed=False, in_mask=None):
    """conv2d(x, weight, padding=1) by bridges_conv3x3 (f32 matrix cores): + bias and ReLU when ``bias`` is given; times
    [mask > 0] when ``mask`` is given; ``transposed``: the input gradient of a layer with ``weight`` [c_in_of_x, c_out, 3, 3];
    ``in_mask`` (shape of x): x counts only where in_mask > 0 (x = the gradient at a ReLU's output)."""
    L = abi.require_gpu()
    x = _vec4(x)
    weight = weight.contiguous()
    n, c_in, H, W = x.shape
    c_out = weight.shape[1] if transposed else weight.shape[0]
    assert (weight.shape[0] if transposed else weight.shape[1]) == c_in and tuple(weight.shape[2:]) == (3, 3) and H == W
    assert not (bias is not None and mask is not None)
    out = torch.empty((n, c_out, H, W), dtype=torch.float32, device=x.device)
    mode = 1 if bias is not None else (2 if mask is not None else 0)
    if mask is not None:
        mask = _vec4(mask)
        assert mask.shape == out.shape
    if in_mask is not None:
        in_mask = _vec4(in_mask)
        assert in_mask.shape == x.shape
    abi.check(L.bridges_conv3x3(_ptr(x), _ptr(in_mask), _ptr(weight), _ptr(bias.contiguous() if bias is not None else None), _ptr(mask),
                                _ptr(out), n, c_in, c_out, W, mode, int(bool(transposed)), _stream()), "bridges_conv3x3")
    return out


_wgrad_scratch = {}


class deferred_wgrad_reduce:
    """``with deferred_wgrad_reduce(tables): loss.backward()`` -- the conv3x3 layers' weight-gradient kernels of that backward
    pass leave their partial sums, and ONE launch (bridges_reduce_jobs) adds them into the parameters' ``.grad`` tensors when the
    block ends, instead of one reduction launch per layer (8 for ConvNet, 18 for the U-Net policy, ~4.6 us each and each in the
    critical path of its layer).  Same arithmetic, same order, same bits.

    Only for a backward pass the caller owns: every parameter is a leaf used ONCE in the graph, its ``.grad`` is None before
    (``zero_grad(set_to_none=True)``) and autograd accumulates into it (``backward()``, not ``autograd.grad``) -- the tensor a
    layer hands autograd as its weight gradient is uninitialised memory until the block ends.  A layer that cannot promise this
    (non-leaf weight, existing ``.grad``, anomaly mode) reduces at once as before.  ``tables``: a ``ReduceTables`` that lives as
    long as any graph captured inside the block (the job table is a captured host-to-device copy)."""
    _active = None

    def __init__(self, tables):
        self.tables, self.jobs = tables, []

    def __enter__(self):
        assert deferred_wgrad_reduce._active is None, "deferred_wgrad_reduce does not nest"
        deferred_wgrad_reduce._active = self
        return self

    def __exit__(self, exc_type, exc, tb):
        deferred_wgrad_reduce._active = None
        if exc_type is None:
            self.tables.launch(self.jobs)
        return False

    @classmethod
    def accepts(cls, weight, bias):
        return (cls._active is not None and weight is not None and bias is not None and weight.is_leaf and bias.is_leaf
                and weight.requires_grad and bias.requires_grad and weight.grad is None and bias.grad is None
                and not torch.is_anomaly_enabled())


def _table_to_device(host, table):
    """A host-built pointer table on the device.  Inside a graph capture: the object's own pinned buffer and device tensor (the
    copy becomes a node that re-reads the pinned buffer at every replay, so nothing else may ever write it).  Eagerly the host
    runs ahead of the GPU -- the next call would overwrite the pinned buffer before this call's copy has run -- so the bytes go
    through the staging ring of ops.upload (a slot is reused only after its copy has completed)."""
    if torch.cuda.is_current_stream_capturing():
        table.copy_(host, non_blocking=True)
        return table
    from . import ops
    return ops.upload(table.device, host.numpy())[0]


class ReduceTables:
    """Host (pinned) and device copies of one backward pass's bridges_reduce_job table."""
    _JOB = np.dtype([("part", "<u8"), ("part_b", "<u8"), ("dw", "<u8"), ("db", "<u8"), ("n_w", "<i4"), ("n_b", "<i4"), ("splits", "<i4"),
                     ("block_start", "<i4")])
    CAPACITY = 64

    def __init__(self, device):
        self.host = torch.empty(self.CAPACITY * self._JOB.itemsize, dtype=torch.uint8).pin_memory()
        self.table = torch.empty(self.host.numel(), dtype=torch.uint8, device=device)
        self.keep = []                                                       # the partial-sum buffers of the last pass

    def launch(self, jobs):
        if not jobs:
            return
        if len(jobs) > self.CAPACITY:
            raise abi.BridgesHipError(f"{len(jobs)} deferred reductions > {self.CAPACITY}")
        L = abi.require_gpu()
        rows = self.host.numpy().view(self._JOB)
        start = 0
        for i, (w, b, scratch, n_w, n_b, splits) in enumerate(jobs):
            gw, gb = w.grad, b.grad
            if gw is None or gb is None or not gw.is_contiguous() or not gb.is_contiguous() or gw.dtype != torch.float32:
                raise abi.BridgesHipError("deferred_wgrad_reduce: a parameter did not receive a contiguous float32 .grad from this pass")
            rows[i] = (scratch.data_ptr(), scratch.data_ptr() + 4 * splits * n_w, gw.data_ptr(), gb.data_ptr(), n_w, n_b, splits, start)
            start += -(-(n_w + n_b) // (256 if splits <= 16 else 16))          # bridges_hip.h: B_j
        self.keep = [j[2] for j in jobs]
        table = _table_to_device(self.host, self.table)
        abi.check(L.bridges_reduce_jobs(_ptr(table), len(jobs), start, _stream()), "bridges_reduce_jobs")


def conv3x3_wgrad(g, x, g_mask=None, owner=None):
    """(dW [c_out, c_in, 3, 3], db [c_out]) of a conv3x3 layer from the gradient g at its output (times [g_mask > 0] when
    given: the layer's ReLU) and its input x (bridges_conv3x3_wgrad: deterministic partial sums + one reduction launch).
    ``owner`` = the layer's (weight, bias) parameters: inside a ``deferred_wgrad_reduce`` block the reduction is left to the
    block's end and the two tensors returned here are placeholders for autograd (see there)."""
    L = abi.require_gpu()
    g, x = _vec4(g), _vec4(x)
    if g_mask is not None:
        g_mask = _vec4(g_mask)
        assert g_mask.shape == g.shape
    n, c_out, H, W = g.shape
    c_in = x.shape[1]
    need = C.c_int64(0)
    abi.check(L.bridges_conv3x3_wgrad_scratch(n, c_in, c_out, W, C.byref(need)), "bridges_conv3x3_wgrad_scratch")
    if owner is not None and deferred_wgrad_reduce.accepts(*owner):
        n_w = c_out * c_in * 9
        scratch = torch.empty(need.value, dtype=torch.float32, device=g.device)       # lives until the block's launch has run
        abi.check(L.bridges_conv3x3_wgrad(_ptr(g), _ptr(g_mask), _ptr(x), None, None, _ptr(scratch), scratch.numel(), n, c_in, c_out, W,
                                          _stream()), "bridges_conv3x3_wgrad")
        deferred_wgrad_reduce._active.jobs.append((owner[0], owner[1], scratch, n_w, c_out, need.value // (n_w + c_out)))
        return (torch.empty((c_out, c_in, 3, 3), dtype=torch.float32, device=g.device),
                torch.empty(c_out, dtype=torch.float32, device=g.device))
    key = (str(g.device), torch.cuda.current_stream().cuda_stream)
    sc = _wgrad_scratch.get(key)
    if sc is None or sc.numel() < need.value:
        sc = _wgrad_scratch[key] = torch.empty(max(need.value, 1 << 20), dtype=torch.float32, device=g.device)
    dw = torch.empty((c_out, c_in, 3, 3), dtype=torch.float32, device=g.device)
    db = torch.empty(c_out, dtype=torch.float32, device=g.device)
    abi.check(L.bridges_conv3x3_wgrad(_ptr(g), _ptr(g_mask), _ptr(x), _ptr(dw), _ptr(db), _ptr(sc), sc.numel(), n, c_in, c_out, W, _stream()),
              "bridges_conv3x3_wgrad")
    return dw, db


def maxpool2(a):
    L = abi.require_gpu()
    a = a.contiguous()
    n, c, H, W = a.shape
    y = torch.empty((n, c, H // 2, W // 2), dtype=torch.float32, device=a.device)
    abi.check(L.bridges_maxpool2(_ptr(a), _ptr(y), n * c, H, W, _stream()), "bridges_maxpool2")
    return y


def maxpool2_relu_backward(a, dy):
    """Gradient at the pre-pool activation a = relu(.) from dy at max_pool2d(a, 2) (first maximum takes it, times [a > 0])."""
    L = abi.require_gpu()
    a, dy = a.contiguous(), dy.contiguous()
    n, c, H, W = a.shape
    g = torch.empty_like(a)
    abi.check(L.bridges_maxpool2_relu_backward(_ptr(a), _ptr(dy), _ptr(g), n * c, H, W, _stream()), "bridges_maxpool2_relu_backward")
    return g


class MaxPool2OfReLUFunction(torch.autograd.Function):
    """max_pool2d(a, 2) of a ReLU OUTPUT a (>= 0): forward bridges_maxpool2, backward the gradient at a with the first maximum
    of a window taking dy (as torch) and zero where a == 0 -- what the ReLU's own backward makes of it anyway, so behind a ReLU
    this is torch's gradient exactly (the U-Net's pool1 / pool2 behind e12 / e22)."""

    @staticmethod
    def forward(ctx, a):
        a = a.contiguous()
        ctx.save_for_backward(a)
        return maxpool2(a)

    @staticmethod
    def backward(ctx, dy):
        a, = ctx.saved_tensors
        return maxpool2_relu_backward(a, dy)


def maxpool2_of_relu_applies(a):
    return a.is_cuda and a.dtype == torch.float32 and a.dim() == 4 and a.shape[2] % 2 == 0 and a.shape[3] % 2 == 0 and torch.is_grad_enabled()


class ConvBlockFunction(torch.autograd.Function):
    """conv3x3 - ReLU - conv3x3 - ReLU - MaxPool2d(2) (cv.py:5-17) with a hand-written forward AND backward: 3 launches
    forward, 7 backward (pool gradient, two weight gradients with their reductions, two input gradients) instead of the
    library's ~14 + ~50."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        a1 = conv3x3(x, w1, bias=b1)
        a2 = conv3x3(a1, w2, bias=b2)
        y = maxpool2(a2)
        ctx.save_for_backward(x, a1, a2, w1, w2)
        ctx.owners = ((w1, b1), (w2, b2))                 # the parameters themselves (deferred_wgrad_reduce writes their .grad)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a1, a2, w1, w2 = ctx.saved_tensors
        g2 = maxpool2_relu_backward(a2, dy)
        dw2, db2 = conv3x3_wgrad(g2, a1, owner=ctx.owners[1])
        g1 = conv3x3(g2, w2, mask=a1, transposed=True)
        dw1, db1 = conv3x3_wgrad(g1, x, owner=ctx.owners[0])
        dx = conv3x3(g1, w1, transposed=True) if ctx.needs_input_grad[0] and x.shape[1] % 16 == 0 else None
        if ctx.needs_input_grad[0] and dx is None:                    # a first layer whose input wants a gradient: the library's
            dx = torch.nn.grad.conv2d_input(x.shape, w1, g1, padding=1)
        return dx, dw1, db1, dw2, db2


class Conv3x3ReLUFunction(torch.autograd.Function):
    """relu(conv2d(x, w, b, padding=1)) with hand-written forward and backward (the U-Net's conv layers, cv.py:138-254): one
    launch forward; backward the weight / bias gradient (two launches) and the input gradient (one), both reading the incoming
    gradient through the ReLU's mask -- no element-wise pass in between."""

    @staticmethod
    def forward(ctx, x, w, b):
        a = conv3x3(x, w, bias=b)
        ctx.save_for_backward(x, a, w)
        ctx.owner = (w, b)
        return a

    @staticmethod
    def backward(ctx, da):
        x, a, w = ctx.saved_tensors
        da = da.contiguous()
        dw, db = conv3x3_wgrad(da, x, g_mask=a, owner=ctx.owner)
        dx = None
        if ctx.needs_input_grad[0]:
            if x.shape[1] % 16 == 0:
                dx = conv3x3(da, w, transposed=True, in_mask=a)
            else:
                dx = torch.nn.grad.conv2d_input(x.shape, w, da * (a > 0), padding=1)
        return dx, dw, db


class BiasAddFunction(torch.autograd.Function):
    """y = x + bias[None, :, None, None] for the output of a bias-free library convolution, with the bias gradient by
    bridges_bias_grad (deterministic, no multi-workgroup reduction): what makes a training step that still holds library
    convolutions (the U-Net's transposed and 1x1 layers) safe to replay from a HIP graph."""

    @staticmethod
    def forward(ctx, x, bias):
        return x + bias.view(1, -1, 1, 1)

    @staticmethod
    def backward(ctx, dy):
        L = abi.require_gpu()
        dy = dy.contiguous()
        n, c = dy.shape[0], dy.shape[1]
        hw = dy[0, 0].numel()
        db = torch.empty(c, dtype=torch.float32, device=dy.device)
        scratch = torch.empty(min(n, 32) * c, dtype=torch.float32, device=dy.device)
        abi.check(L.bridges_bias_grad(_ptr(dy), _ptr(db), _ptr(scratch), scratch.numel(), n, c, hw, _stream()), "bridges_bias_grad")
        return dy, db


class UpConv2x2Function(torch.autograd.Function):
    """ConvTranspose2d(kernel 2, stride 2) + bias of the U-Net decoder, forward by bridges_upconv2x2 (the inference kernel),
    backward by bridges_upconv2x2_backward: input gradient, weight and bias gradient as three launches (deterministic partial
    sums) where the library ran two convolution kernels, an implicit-GEMM weight gradient between layout transposes and a fill."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        x = _vec4(x)
        ctx.save_for_backward(x, weight)
        ctx.owner = (weight, bias)
        return upconv2x2(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        L = abi.require_gpu()
        x, weight = ctx.saved_tensors
        dy = _vec4(dy)
        w = weight.contiguous()
        n, c_in, H, W = x.shape
        c_out = w.shape[1]
        need = C.c_int64(0)
        abi.check(L.bridges_upconv2x2_backward_scratch(n, c_in, c_out, H, W, C.byref(need)), "bridges_upconv2x2_backward_scratch")
        scratch = torch.empty(need.value, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(w)
        db = torch.empty(c_out, dtype=torch.float32, device=x.device)
        defer = deferred_wgrad_reduce.accepts(*ctx.owner)            # placeholders for autograd, one reduction launch at the block's end
        abi.check(L.bridges_upconv2x2_backward(_ptr(x), _ptr(dy), _ptr(w), _ptr(dx), None if defer else _ptr(dw), None if defer else _ptr(db),
                                               _ptr(scratch), scratch.numel(), n, c_in, c_out, H, W, _stream()), "bridges_upconv2x2_backward")
        if defer:
            n_w = c_in * c_out * 4
            deferred_wgrad_reduce._active.jobs.append((ctx.owner[0], ctx.owner[1], scratch, n_w, c_out, need.value // (n_w + c_out)))
        return dx, dw, db


def upconv2x2_train_applies(x, up):
    return upconv2x2_applies(x, up) and (x.shape[2] * x.shape[3]) % 64 == 0


class Conv1x1O1Function(torch.autograd.Function):
    """Conv2d(c_in, 1, kernel_size=1) (the U-Net's outconv for one class): forward one pass over x, backward one pass writing dx
    and the partial sums of dw / db, then their fixed-order reduction."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        L = abi.require_gpu()
        x = _vec4(x)
        n, c_in, H, W = x.shape
        w = weight.reshape(-1).contiguous()
        y = torch.empty((n, 1, H, W), dtype=torch.float32, device=x.device)
        abi.check(L.bridges_conv1x1_o1_forward(_ptr(x), _ptr(w), _ptr(bias.contiguous()), _ptr(y), n, c_in, H * W, _stream()), "bridges_conv1x1_o1_forward")
        ctx.save_for_backward(x, w)
        ctx.w_shape = weight.shape
        ctx.owner = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        L = abi.require_gpu()
        x, w = ctx.saved_tensors
        dy = _vec4(dy)
        n, c_in, H, W = x.shape
        S = min(256, max(1, -(-(n * H * W // 4) // 256)))
        scratch = torch.empty(S * (c_in + 1), dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        dw = torch.empty(c_in, dtype=torch.float32, device=x.device)
        db = torch.empty(1, dtype=torch.float32, device=x.device)
        defer = deferred_wgrad_reduce.accepts(*ctx.owner)
        abi.check(L.bridges_conv1x1_o1_backward(_ptr(x), _ptr(dy), _ptr(w), _ptr(dx), None if defer else _ptr(dw), None if defer else _ptr(db),
                                                _ptr(scratch), scratch.numel(), n, c_in, H * W, _stream()), "bridges_conv1x1_o1_backward")
        if defer:
            deferred_wgrad_reduce._active.jobs.append((ctx.owner[0], ctx.owner[1], scratch, c_in, 1, S))
        return dx, dw.view(ctx.w_shape), db


def conv1x1_o1_applies(x, conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.out_channels == 1 and tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (0, 0) and conv.groups == 1 and conv.bias is not None and conv.in_channels <= 32 and x.dim() == 4
            and (x.shape[2] * x.shape[3]) % 4 == 0 and x.dtype == torch.float32 and x.is_cuda)


def conv_bias_train(module, x):
    """module(x) for the U-Net's transposed / 1x1 convolutions in a training pass on the GPU: the hand-written pair where it
    applies (UpConv2x2Function, Conv1x1O1Function); else the library's convolution without its bias and the bias through
    BiasAddFunction."""
    if not (torch.is_grad_enabled() and x.is_cuda and x.dtype == torch.float32 and module.bias is not None):
        return module(x)
    if isinstance(module, torch.nn.ConvTranspose2d) and upconv2x2_train_applies(x, module):
        return UpConv2x2Function.apply(x, module.weight, module.bias)
    if conv1x1_o1_applies(x, module):
        return Conv1x1O1Function.apply(x, module.weight, module.bias)
    if isinstance(module, torch.nn.ConvTranspose2d):
        y = torch.nn.functional.conv_transpose2d(x, module.weight, None, module.stride, module.padding, module.output_padding,
                                                 module.groups, module.dilation)
    else:
        y = torch.nn.functional.conv2d(x, module.weight, None, module.stride, module.padding, module.dilation, module.groups)
    return BiasAddFunction.apply(y, module.bias)


def conv3x3_relu_train(conv, x):
    """relu(conv(x)) for a training pass: the hand-written Function where it applies, else the library."""
    if (torch.is_grad_enabled() and conv3x3_supported(x, conv.out_channels) and tuple(conv.kernel_size) == (3, 3)
            and tuple(conv.padding) == (1, 1) and tuple(conv.stride) == (1, 1) and conv.groups == 1 and conv.bias is not None):
        return Conv3x3ReLUFunction.apply(x, conv.weight, conv.bias)
    return torch.relu(conv(x))


class FlatParameters:
    """All parameters and float buffers of a module re-pointed into ONE contiguous float32 device buffer, so the
    Polyak update of a 6.4 M-parameter SuccessorMLP is a single launch instead of one per state_dict key."""

    def __init__(self, module):
        tensors = [p for p in module.parameters()] + [b for b in module.buffers() if b.dtype == torch.float32]
        offsets, n = [], 0
        for t in tensors:
            offsets.append(n)
            n += (t.numel() + 3) // 4 * 4                  # keep every tensor 16-byte aligned
        dev = tensors[0].device
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        for t, off in zip(tensors, offsets):
            view = self.flat[off:off + t.numel()].view_as(t)
            view.copy_(t.data)
            t.data = view
        self.module = module


class MultiTensorAdam:
    """``optimizer.step()`` of a plain torch.optim.Adam (one parameter group; amsgrad, weight decay, maximize off) over all its
    float32 GPU parameters as ONE launch of 1024-element chunks (bridges_adam_multi) -- torch's fused multi-tensor launch cuts
    the 34-60 tensors of a conv Q-network into 64 k-element chunks, ~40 workgroups, 40-63 us per step.  The optimiser's own state
    tensors are the moments and receive the step count, so ``optimizer.step()`` / ``state_dict()`` carry on from here at any
    time.  ``step()`` reads the gradients' addresses when it is called: inside a graph capture they are the capture's.
    Raises ValueError when the optimiser is not of that kind or has not taken its first step yet (no state to adopt)."""
    CHUNK = 1024
    _SLOT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("step", "<u8"), ("n", "<i8")])

    def __init__(self, optimizer):
        if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1:
            raise ValueError("not a plain torch.optim.Adam with one parameter group")
        g = optimizer.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("weight_decay") or g.get("differentiable"):
            raise ValueError("amsgrad / maximize / weight decay / differentiable are not covered")
        self.opt, self.params = optimizer, [p for p in g["params"] if p.requires_grad]
        steps = set()
        for p in self.params:
            st = optimizer.state.get(p)
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and st and torch.is_tensor(st.get("step")) and st["step"].is_cuda
                    and st["step"].dtype == torch.float32 and st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous()):
                raise ValueError("parameters must be contiguous float32 GPU tensors with an initialised fused-Adam state")
            steps.add(float(st["step"]))
        if len(steps) != 1:
            raise ValueError("the parameters' step counts differ")
        dev = self.params[0].device
        self.step_count = torch.full((), steps.pop(), dtype=torch.float32, device=dev)      # updates done so far
        slot_of, off = [], []
        for i, p in enumerate(self.params):
            n_chunks = -(-p.numel() // self.CHUNK)
            slot_of += [i] * n_chunks
            off += list(range(n_chunks))
        self.n_chunks = len(slot_of)
        self.chunk_slot = torch.tensor(slot_of, dtype=torch.int32).to(dev)
        self.chunk_off = torch.tensor(off, dtype=torch.int32).to(dev)
        self.host = torch.empty(len(self.params) * self._SLOT.itemsize, dtype=torch.uint8).pin_memory()
        self.table = torch.empty(self.host.numel(), dtype=torch.uint8, device=dev)
        self.hyper = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))

    def check_hyperparameters(self):
        g = self.opt.param_groups[0]
        now = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
        if now != self.hyper:
            raise abi.BridgesHipError(f"the optimiser's hyper-parameters changed ({self.hyper} -> {now}): build a new MultiTensorAdam")

    def step(self):
        L = abi.require_gpu()
        rows = self.host.numpy().view(self._SLOT)
        for i, p in enumerate(self.params):
            if p.grad is None or not p.grad.is_contiguous() or p.grad.dtype != torch.float32:
                raise abi.BridgesHipError("MultiTensorAdam.step: every parameter needs a contiguous float32 gradient")
            st = self.opt.state[p]
            rows[i] = (p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr(), p.numel())
        table = _table_to_device(self.host, self.table)
        lr, b1, b2, eps = self.hyper
        abi.check(L.bridges_adam_multi(_ptr(table), len(self.params), _ptr(self.chunk_slot), _ptr(self.chunk_off), self.n_chunks,
                                       _ptr(self.step_count), lr, b1, b2, eps, _stream()), "bridges_adam_multi")
        self.step_count.add_(1.0)
