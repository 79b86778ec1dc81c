"""One optimiser step's forward + losses + backward of SuccessorMLP at replay-batch size as ~20 HIP launches on the
f32 matrix cores (csrc/mlp_kernels.hip; robotoddler/models/cv.py:76-105, train_policy_net
robotoddler/training/successor_dqn.py:157-235).  The parameters stay the module's own tensors, the gradients land in
their ``.grad`` -- the optimiser (torch's fused Adam) is untouched.  No CPU fallback: abi.require_gpu() raises without
the HIP library."""
import torch

from . import abi
from .ops import _ptr, _stream


def linear_forward(x, weight, bias, relu, out=None, ws=None):
    """act(x @ weight.T + bias) for x [rows, K] with rows % 32 == 0 (bridges_linear_forward)."""
    L = abi.require_gpu()
    rows, K = x.shape
    N = weight.shape[0]
    for t in (x, weight, bias):
        assert t.dtype == torch.float32 and t.is_contiguous() and t.is_cuda
    out = torch.empty((rows, N), dtype=torch.float32, device=x.device) if out is None else out
    ws = torch.empty(1 << 20, dtype=torch.float32, device=x.device) if ws is None else ws
    abi.check(L.bridges_linear_forward(rows, K, N, _ptr(x), _ptr(weight), _ptr(bias), int(bool(relu)), _ptr(out), _ptr(ws),
                                       ws.numel(), _stream()), "bridges_linear_forward")
    return out


def linear_backward(dz, a_in, weight, act_below=None, need_input_grad=True, ws=None):
    """-> (dW, db, dz_below or None) of y = a_in @ weight.T + b from dz = dL/dy (bridges_linear_backward);
    dz_below is masked by act_below > 0 when that is given (the ReLU whose output a_in is)."""
    L = abi.require_gpu()
    rows, N = dz.shape
    K = a_in.shape[1]
    dW, db = torch.empty_like(weight), torch.empty(N, dtype=torch.float32, device=dz.device)
    below = torch.empty((rows, K), dtype=torch.float32, device=dz.device) if need_input_grad else None
    ws = torch.empty(1 << 20, dtype=torch.float32, device=dz.device) if ws is None else ws
    abi.check(L.bridges_linear_backward(rows, K, N, _ptr(dz), _ptr(a_in), _ptr(weight), _ptr(dW), _ptr(db), _ptr(act_below),
                                        _ptr(below), _ptr(ws), ws.numel(), _stream()), "bridges_linear_backward")
    return dW, db, below


class FusedSuccessorStep:
    """Static buffers + launch sequence of one SuccessorMLP training step for a fixed batch size.

    ``launch`` reads replay batch ``counter`` of the per-call arrays (row counter * batch + b), leaves the loss in
    ``losses[counter]``, the gradients in the parameters' ``.grad`` and increments ``counter`` -- every argument is a
    device tensor and nothing synchronises, so the sequence can be captured in a HIP graph together with the
    optimiser step."""

    WS_FLOATS = 4 << 20

    def __init__(self, net, batch, use_q, use_sf):
        self.L = abi.require_gpu()
        self.linears = [m for m in net.mlp.layers if isinstance(m, torch.nn.Linear)]
        self.px = int(net.img_size[0]) * int(net.img_size[1])
        dims = [self.linears[0].in_features] + [lin.out_features for lin in self.linears]
        self.nf = dims[0] - 4 * self.px
        if self.nf < 0 or dims[-1] != 2 * self.px + 2 * self.nf:
            raise ValueError("not a SuccessorMLP over 4 image channels + binary features")
        for a, lin in zip(dims[:-1], self.linears):
            assert lin.in_features == a
        self.batch, self.rows = int(batch), 32 * ((int(batch) + 31) // 32)
        self.use_q, self.use_sf = bool(use_q), bool(use_sf)
        dev = self.linears[0].weight.device
        z = lambda *s: torch.zeros(s, dtype=torch.float32, device=dev)
        self.acts = [z(self.rows, d) for d in dims]                  # acts[0] = input rows, acts[-1] = network output
        self.dz = [z(self.rows, d) for d in dims[1:]]                # gradient at the pre-activation of every layer
        self.ws = torch.empty(self.WS_FLOATS, dtype=torch.float32, device=dev)
        self.loss_rows, self.q = z(self.rows), z(self.rows)
        # the gradient tensors the launches write: referenced here as well, so a later zero_grad(set_to_none=True) cannot
        # hand their memory to someone else while a captured graph still writes to it
        self._grads = []
        for lin in self.linears:
            for p in (lin.weight, lin.bias):
                assert p.dtype == torch.float32 and p.is_contiguous()
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                assert p.grad.is_contiguous()
                self._grads.append(p.grad)

    def launch(self, counter, block_all, action_all, binary_all, reward, obstacle, q_target_all, sf_target_all, losses):
        L, rows, px, nf, B = self.L, self.rows, self.px, self.nf, self.batch
        st = _stream()
        for t in (block_all, action_all, binary_all, reward, obstacle):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert counter.dtype == torch.int64 and losses.dtype == torch.float32
        abi.check(L.bridges_mlp_input(B, rows, px, nf, _ptr(counter), _ptr(block_all), _ptr(action_all), _ptr(binary_all),
                                      _ptr(reward), _ptr(obstacle), _ptr(self.acts[0]), st), "bridges_mlp_input")
        last = len(self.linears) - 1
        for l, lin in enumerate(self.linears):
            abi.check(L.bridges_linear_forward(rows, lin.in_features, lin.out_features, _ptr(self.acts[l]), _ptr(lin.weight),
                                               _ptr(lin.bias), int(l < last), _ptr(self.acts[l + 1]), _ptr(self.ws),
                                               self.ws.numel(), st), "bridges_linear_forward")
        abi.check(L.bridges_successor_loss(B, rows, px, nf, _ptr(self.acts[-1]), _ptr(reward), _ptr(counter),
                                           _ptr(q_target_all) if self.use_q else None,
                                           _ptr(sf_target_all) if self.use_sf else None, int(self.use_q), int(self.use_sf),
                                           _ptr(self.dz[-1]), _ptr(self.loss_rows), _ptr(self.q), _ptr(losses),
                                           losses.numel(), _ptr(counter), st), "bridges_successor_loss")
        for l in range(last, -1, -1):
            lin = self.linears[l]
            abi.check(L.bridges_linear_backward(rows, lin.in_features, lin.out_features, _ptr(self.dz[l]), _ptr(self.acts[l]),
                                                _ptr(lin.weight), _ptr(lin.weight.grad), _ptr(lin.bias.grad),
                                                _ptr(self.acts[l]) if l > 0 else None,
                                                _ptr(self.dz[l - 1]) if l > 0 else None, _ptr(self.ws), self.ws.numel(), st),
                      "bridges_linear_backward")
