"""Q-networks of the successor-feature DQN (robotoddler/models/cv.py:5-271 of the reference).

Same module / parameter names (state_dicts are interchangeable) and the same forward contract
``net(block_f, binary_f, action_f, reward_f, obstacle_f) -> (q, succ_block_f, succ_binary_f)``.
ConvNet implements the 5-argument forward the training loop calls (reference cv.py:67-73, commented out at HEAD,
where only the 2-argument form used by Policy.SFStability is live); both forms are supported here.
The networks stay in PyTorch-ROCm (MIOpen / hipBLASLt); hand-written HIP around them: the target construction and the
soft update (bridges_td_target, bridges_soft_update), the bit-packed first layer and head of SuccessorMLP's acting forward,
and -- for the inference passes (acting, targets: no autograd) of the conv nets -- the epilogue of every convolution:
``+ bias -> ReLU [-> MaxPool2d(2)]`` in ONE pass over the activations (bridges_bias_relu, bridges_bias_relu_pool2) instead
of torch's three, bit-identical to the module's own forward.
"""
import os

import torch
from torch import nn
import torch.nn.functional as F

# The library convolutions left on this path are the inference passes of the 32- / 64- / 128-channel layers (acting, targets:
# thousands of rows).  For those MIOpen's immediate-mode choice on gfx950 is often its NHWC implicit-GEMM forward solver wrapped
# in two layout transposes -- measured against its own Winograd kernels on the same layers (tools/train_throughput.py, U-Net
# policy at 4096 envs): acting 14.7 -> 11.8 ms, targets 12.3 -> 9.2 ms per lock-step with that one solver off.  A default only:
# an explicit setting of the variable wins.  (It must be in the environment before the process's first convolution.)
os.environ.setdefault("MIOPEN_DEBUG_CONV_IMPLICIT_GEMM_ASM_FWD_GTC_XDLOPS_NHWC", "0")


def _fused_inference(x):
    """True when the fused epilogues apply: no autograd graph is being recorded, float32 tensors on the GPU."""
    return (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32


def _conv_relu(conv, x, pool=False):
    """relu(conv(x)) [then 2x2 max-pool] with the bias / ReLU / pool epilogue as one HIP pass."""
    from bridges_hip import dqn_ops
    if dqn_ops.conv3x3_relu_o16_applies(x, conv):           # the 64-wide, 16-channel layers: hand-written MFMA kernel
        return dqn_ops.conv3x3_relu_o16(x, conv.weight, conv.bias, pool)
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups).contiguous()
    if pool:
        return dqn_ops.bias_relu_pool2(y, conv.bias)
    return dqn_ops.bias_relu_(y, conv.bias)


def _conv_pair(cin, cout):
    return [nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.ReLU(),
            nn.Conv2d(cout, cout, kernel_size=3, padding=1), nn.ReLU()]


class ConvBlock(nn.Module):
    """conv3x3-relu-conv3x3-relu-maxpool2 (cv.py:5-17)."""

    def __init__(self, in_c, out_c):
        super().__init__()
        self.layers = nn.Sequential(*_conv_pair(in_c, out_c), nn.MaxPool2d((2, 2)))

    def forward(self, inputs):
        if _fused_inference(inputs) and inputs.shape[-1] % 8 == 0 and inputs.shape[-2] % 2 == 0:
            return _conv_relu(self.layers[2], _conv_relu(self.layers[0], inputs), pool=True)
        if torch.is_grad_enabled() and inputs.is_cuda and inputs.dtype == torch.float32:
            # training pass (train_policy_net): forward and backward of the whole block on the hand-written kernels
            from bridges_hip import dqn_ops
            c1, c2 = self.layers[0], self.layers[2]
            if dqn_ops.conv3x3_supported(inputs, c1.out_channels) and c1.out_channels == c2.out_channels == c2.in_channels:
                return dqn_ops.ConvBlockFunction.apply(inputs, c1.weight, c1.bias, c2.weight, c2.bias)
        return self.layers(inputs)


class MLP(nn.Module):
    """Linear-ReLU stack with a linear head (cv.py:20-38)."""

    def __init__(self, in_d, out_d, hidden_dim=None):
        super().__init__()
        dims = [in_d] + list(hidden_dim if hidden_dim is not None else [64])
        mods = []
        for a, b in zip(dims[:-1], dims[1:]):
            mods += [nn.Linear(a, b), nn.ReLU()]
        mods.append(nn.Linear(dims[-1], out_d))
        self.layers = nn.Sequential(*mods)

    def forward(self, inputs):
        return self.layers(inputs)


class ConvNet(nn.Module):
    """Four ConvBlocks (16/32/64/128) + MLP head -> q-value and binary successor features (cv.py:41-73)."""

    def __init__(self, in_channels=4, img_size=(512, 512), num_features=6):
        super().__init__()
        self.layers = nn.Sequential(ConvBlock(in_channels, 16), ConvBlock(16, 32), ConvBlock(32, 64), ConvBlock(64, 128))
        self.bottleneck_size = 128 * (img_size[0] // 16) * (img_size[1] // 16)
        self.num_features = num_features
        self.mlp = MLP(self.bottleneck_size + num_features, 2 * num_features + 1)

    def forward(self, block_features, *rest):
        if len(rest) == 1:                                   # (block, action): Policy.SFStability (cv.py:61-65)
            x = torch.cat([block_features, rest[0]], dim=1)
            return self.mlp(self.layers(x).reshape(-1, self.bottleneck_size))
        binary_features, action_features, reward_features, obstacle_features = rest          # cv.py:67-73
        x = torch.cat([block_features, action_features, reward_features, obstacle_features], dim=1)
        x = self.layers(x).reshape(-1, self.bottleneck_size)
        x = self.mlp(torch.cat([x, binary_features], dim=1))
        q_values = x[:, 0]
        succ_binary_features = x[:, 1:].reshape(-1, 2, binary_features.shape[1])
        return q_values, None, succ_binary_features


class SuccessorMLP(nn.Module):
    """MLP with a bottleneck predicting successor images/features; q = sum(softmax(psi)[:,1] * reward map)
    (cv.py:76-105)."""

    def __init__(self, in_channels=4, img_size=(512, 512), num_features=6, hidden_dims=None):
        super().__init__()
        self.img_size = img_size
        px = img_size[0] * img_size[1]
        self.mlp = MLP(in_channels * px + num_features, 2 * px + 2 * num_features,
                       hidden_dims if hidden_dims is not None else [128, 64, 128])

    def forward(self, block_features, binary_features, action_features, reward_features, obstacle_features):
        n = block_features.shape[0]
        x = torch.cat([block_features, action_features, reward_features, obstacle_features], dim=1).reshape(n, -1)
        x = self.mlp(torch.cat([x, binary_features], dim=1))
        img_dim = 2 * self.img_size[0] * self.img_size[1]
        succ_block_features = x[:, :img_dim].reshape(-1, 2, *self.img_size)
        succ_binary_features = x[:, img_dim:].reshape(-1, 2, binary_features.shape[1])
        q_values = torch.sum(succ_block_features.softmax(dim=1)[:, 1] * reward_features.squeeze(1), dim=(-1, -2))
        return q_values, succ_block_features, succ_binary_features


    @torch.no_grad()
    def q_values_factored(self, block_features, binary_features, action_features, row_env, reward_features,
                          obstacle_features):
        """q of ``forward`` for n candidate rows that share per-environment inputs, without materialising the
        [n, 4*px+f] input or the [n, 2*px+2f] output (acting needs only q):
          * first layer: W1 [block | action | reward | obstacle | binary] = (per env: block, binary; constant: reward,
            obstacle, bias) + (per row: action) -- a quarter of the multiply-adds, no 4-image concatenation;
          * head: softmax over the two successor channels, channel 1 = sigmoid(psi1 - psi0), so one px-wide product with
            the row difference of the last layer replaces the 2*px+2f-wide one.
        Same function as ``forward(...)[0]`` up to float32 summation order.
        block_features [E,H,W], binary_features [E,f], action_features [n,H,W], row_env [n] (env of each row),
        reward_features / obstacle_features [H,W] (or any shape with H*W elements)."""
        px = self.img_size[0] * self.img_size[1]
        E, n = block_features.shape[0], action_features.shape[0]
        W1 = self.first_layer().weight
        base = block_features.reshape(E, px) @ W1[:, :px].T + self.first_layer_env_terms(binary_features, reward_features,
                                                                                         obstacle_features)
        h = torch.addmm(base.index_select(0, row_env), action_features.reshape(n, px), W1[:, px:2 * px].T)
        return self.q_from_first_layer(h, reward_features)

    def first_layer(self):
        return next(m for m in self.mlp.layers if isinstance(m, nn.Linear))

    @torch.no_grad()
    def first_layer_env_terms(self, binary_features, reward_features, obstacle_features):
        """[E, hidden] part of the first layer that does not depend on the block / action images: binary features,
        reward map, obstacle raster and the bias."""
        px = self.img_size[0] * self.img_size[1]
        lin = self.first_layer()
        W1 = lin.weight
        const = W1[:, 2 * px:3 * px] @ reward_features.reshape(px) + W1[:, 3 * px:4 * px] @ obstacle_features.reshape(px) + lin.bias
        return binary_features @ W1[:, 4 * px:].T + const

    @torch.no_grad()
    def first_layer_stable_table(self, reward_obstacle):
        """first_layer_env_terms for the binary vectors the simulator produces, [stable, 0, 0, 0, 0, 0] (the five collision
        flags are 0 without pybullet, assembly_env.py:311-312): a [2, hidden] table, row s = the terms of an env with stable = s.
        ``reward_obstacle`` = the flattened reward map followed by the flattened obstacle raster ([2 px])."""
        px = self.img_size[0] * self.img_size[1]
        lin = self.first_layer()
        const = torch.addmv(lin.bias, lin.weight[:, 2 * px:4 * px], reward_obstacle)
        return torch.stack([const, const + lin.weight[:, 4 * px]])

    @torch.no_grad()
    def sf0_from_first_layer(self, h_pre):
        """Channel 0 of the successor block features (``forward(...)[1][:, 0]`` flattened to [n, px]) from the
        pre-activation of the first layer: the remaining hidden layers and the first px rows of the output layer -- what the
        successor-feature target of train_policy_net reads of the target net (successor_dqn.py:206-213), without the
        [n, 4*px+f] input, the other 2*px+2f-px outputs or the softmax of the module forward."""
        lin = [m for m in self.mlp.layers if isinstance(m, nn.Linear)]
        px = self.img_size[0] * self.img_size[1]
        h = self._middle_layers(h_pre, lin)
        return torch.addmm(lin[-1].bias[:px], h, lin[-1].weight[:px].T)

    @staticmethod
    def _middle_layers(h_pre, lin):
        """relu(h_pre) through the Linear + ReLU layers between the first and the last (inference)."""
        n = h_pre.shape[0]
        if h_pre.is_cuda and len(lin) > 2:
            from bridges_hip import mlp_ops
            h = mlp_ops.mid_rows(h_pre, lin[1:-1])                  # the whole stack in one launch where the library has it
            if h is not None:
                return h
            # the row count changes with every call, and a row count the GEMM library has not seen costs 77 us of host time per
            # call against 20 us for one it has (tools/addmm_host_cost.py): the layers run on the rows padded to a multiple
            # of 512 (zero rows, sliced off below), so a run meets a handful of shapes instead of thousands
            n_pad = -(-n // 512) * 512
            h = torch.empty((n_pad, h_pre.shape[1]), dtype=h_pre.dtype, device=h_pre.device)
            torch.clamp(h_pre, min=0, out=h[:n])                                 # relu into the padded buffer
            h[n:].zero_()
            for layer in lin[1:-1]:
                h = torch._addmm_activation(layer.bias, h, layer.weight.T)       # bias + ReLU in the library GEMM's epilogue
            return h[:n]
        h = F.relu(h_pre)
        for layer in lin[1:-1]:
            h = F.relu(layer(h))
        return h

    @torch.no_grad()
    def q_from_first_layer(self, h_pre, reward_features, head=None, fused_head=None):
        """q from the pre-activation of the first layer ([n, hidden]): the remaining layers and the factored head.
        ``head(d, w) -> sum_j w[j] * sigmoid(d[:, j])`` may be supplied as a fused operator, or -- for a last hidden width of
        256 -- ``fused_head(h, Wd, bd, w) -> sum_j w[j] * sigmoid(h . Wd[j] + bd[j])``, which never stores the [n, px] product."""
        lin = [m for m in self.mlp.layers if isinstance(m, nn.Linear)]
        px = self.img_size[0] * self.img_size[1]
        h = self._middle_layers(h_pre, lin)
        Wo, bo = lin[-1].weight, lin[-1].bias
        if fused_head is not None and h.shape[1] == 256:
            return fused_head(h, Wo[px:2 * px] - Wo[:px], bo[px:2 * px] - bo[:px], reward_features.reshape(px))
        d = torch.addmm(bo[px:2 * px] - bo[:px], h, (Wo[px:2 * px] - Wo[:px]).T)
        if head is not None:
            return head(d, reward_features.reshape(px))
        return (torch.sigmoid(d) * reward_features.reshape(1, px)).sum(dim=1)


class UNet(nn.Module):
    """Three-level U-Net on the 4 stacked rasters (cv.py:138-254; the deeper levels are disabled there too)."""

    def __init__(self, n_class):
        super().__init__()
        self.n_class = n_class
        self.e11 = nn.Conv2d(4, 16, kernel_size=3, padding=1)
        self.e12 = nn.Conv2d(16, 16, kernel_size=3, padding=1)
        self.pool1 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.e21 = nn.Conv2d(16, 32, kernel_size=3, padding=1)
        self.e22 = nn.Conv2d(32, 32, kernel_size=3, padding=1)
        self.pool2 = nn.MaxPool2d(kernel_size=2, stride=2)
        self.e31 = nn.Conv2d(32, 64, kernel_size=3, padding=1)
        self.e32 = nn.Conv2d(64, 64, kernel_size=3, padding=1)
        self.upconv3 = nn.ConvTranspose2d(64, 32, kernel_size=2, stride=2)
        self.d31 = nn.Conv2d(64, 32, kernel_size=3, padding=1)
        self.d32 = nn.Conv2d(32, 32, kernel_size=3, padding=1)
        self.upconv4 = nn.ConvTranspose2d(32, 16, kernel_size=2, stride=2)
        self.d41 = nn.Conv2d(32, 16, kernel_size=3, padding=1)
        self.d42 = nn.Conv2d(16, 16, kernel_size=3, padding=1)
        self.outconv = nn.Conv2d(16, n_class, kernel_size=1)

    def forward(self, block_features, binary_features, action_features, reward_features, obstacle_features):
        x = torch.cat([block_features, action_features, reward_features, obstacle_features], dim=1)
        from bridges_hip import dqn_ops
        cr = _conv_relu if (_fused_inference(x) and x.shape[-1] % 16 == 0) else (
            dqn_ops.conv3x3_relu_train if (torch.is_grad_enabled() and x.is_cuda) else (lambda conv, t: F.relu(conv(t))))
        if _fused_inference(x) and dqn_ops.conv3x3_relu_o16_applies(x, self.e11):
            # 64x64 inference: the four 16-channel layers on the hand-written kernel with their neighbours folded in -- the
            # first pooling comes out of e12 together with the skip tensor, d41 reads (upconv4, skip) without the
            # concatenation, the 1x1 outconv to one channel is d42's epilogue
            s1, p1 = dqn_ops.conv3x3_relu_o16(cr(self.e11, x), self.e12.weight, self.e12.bias, both=True)
            s2 = cr(self.e22, cr(self.e21, p1))
            b = cr(self.e32, cr(self.e31, self.pool2(s2)))
            up = lambda m, t: dqn_ops.upconv2x2(t, m.weight, m.bias) if dqn_ops.upconv2x2_applies(t, m) else m(t)
            u = torch.cat([up(self.upconv3, b), s2], dim=1)
            u = cr(self.d32, cr(self.d31, u))
            u = dqn_ops.conv3x3_relu_o16(up(self.upconv4, u), self.d41.weight, self.d41.bias, x2=s1)
            if self.n_class == 1:
                return dqn_ops.conv3x3_relu_o16(u, self.d42.weight, self.d42.bias, proj=(self.outconv.weight, self.outconv.bias))
            out = self.outconv(cr(self.d42, u))
        else:
            lib = dqn_ops.conv_bias_train if x.is_cuda else (lambda m, t: m(t))      # transposed / 1x1 layers: hand-written pairs
            # (pool1 / pool2 sit behind a ReLU: the hand-written pooling pair is exact there)
            pool = lambda m, t: dqn_ops.MaxPool2OfReLUFunction.apply(t) if dqn_ops.maxpool2_of_relu_applies(t) else m(t)
            s1 = cr(self.e12, cr(self.e11, x))
            s2 = cr(self.e22, cr(self.e21, pool(self.pool1, s1)))
            b = cr(self.e32, cr(self.e31, pool(self.pool2, s2)))
            u = torch.cat([lib(self.upconv3, b), s2], dim=1)
            u = cr(self.d32, cr(self.d31, u))
            u = torch.cat([lib(self.upconv4, u), s1], dim=1)
            u = cr(self.d42, cr(self.d41, u))
            out = lib(self.outconv, u)
        if self.n_class == 2:
            out = out.softmax(dim=1)[:, 1]
        return out


class Policy(nn.Module):
    """U-Net successor image + ConvNet stability head (cv.py:257-271):
    q = sum(psi * w) * (1 - exp(-10 s)) - exp(-10 s)."""

    def __init__(self):
        super().__init__()
        self.SFImage = UNet(1)
        self.SFStability = ConvNet(in_channels=2, img_size=(64, 64), num_features=0)

    def forward(self, block_features, binary_features, action_features, reward_features, obstacle_features):
        succ_block_features = self.SFImage(block_features, binary_features, action_features, reward_features, obstacle_features)
        stability = torch.sigmoid(self.SFStability(block_features, action_features))
        gate = torch.exp(-10 * stability.squeeze())
        q_values = torch.sum(succ_block_features[:, 0] * reward_features.squeeze(1), dim=(-1, -2)) * (1 - gate) - gate
        return q_values, succ_block_features, stability
