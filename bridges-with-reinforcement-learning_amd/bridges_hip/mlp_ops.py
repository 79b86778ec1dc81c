"""One optimiser step's forward + losses + backward of SuccessorMLP at replay-batch size as ~20 HIP launches on the
f32 matrix cores (csrc/mlp_kernels.hip; robotoddler/models/cv.py:76-105, train_policy_net
robotoddler/training/successor_dqn.py:157-235).  The parameters stay the module's own tensors, the gradients land in
their ``.grad`` -- the optimiser (torch's fused Adam) is untouched.  No CPU fallback: abi.require_gpu() raises without
the HIP library."""
import ctypes as C

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
                                       ws.numel(), None, _stream()), "bridges_linear_forward")
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
                                        _ptr(below), _ptr(ws), ws.numel(), None, 0, _stream()), "bridges_linear_backward")
    return dW, db, below


def mid_rows(h_pre, linears):
    """The Linear + ReLU layers ``linears`` (the reference's 256-128-64-128-256 stack) applied to relu(h_pre) for every row, in
    one launch (bridges_mlp_mid_rows); returns None when the library is not built for the stack (the caller then runs the
    layers one by one).  h_pre [n, 256] float32 with contiguous rows."""
    L = abi.require_gpu()
    n = len(linears)
    if n == 0 or h_pre.dtype != torch.float32 or h_pre.dim() != 2 or h_pre.stride(1) != 1:
        return None
    dims = (C.c_int32 * (n + 1))(linears[0].in_features, *[l.out_features for l in linears])
    if h_pre.shape[1] != linears[0].in_features or not L.bridges_mlp_mid_supported(32, n, dims):
        return None
    if h_pre.stride(0) % 4 or h_pre.data_ptr() % 16 or any(l.weight.data_ptr() % 16 or not l.weight.is_contiguous() for l in linears):
        return None
    VP = C.c_void_p * n
    out = torch.empty((h_pre.shape[0], linears[-1].out_features), dtype=torch.float32, device=h_pre.device)
    mid = torch.empty((h_pre.shape[0], 64), dtype=torch.float32, device=h_pre.device)
    abi.check(L.bridges_mlp_mid_rows(h_pre.shape[0], n, dims, VP(*[l.weight.data_ptr() for l in linears]),
                                     VP(*[l.bias.data_ptr() for l in linears]), _ptr(h_pre), h_pre.stride(0), _ptr(out), out.stride(0),
                                     _ptr(mid), _stream()), "bridges_mlp_mid_rows")
    return out


class FusedSuccessorStep:
    """Static buffers + launch sequence of one SuccessorMLP training step for a fixed batch size.

    ``launch`` reads replay batch ``counter`` of the per-call arrays (row counter * batch + b), leaves the loss in
    ``losses[counter]``, the gradients in the parameters' ``.grad`` and increments ``counter`` -- every argument is a
    device tensor and nothing synchronises, so the sequence can be captured in a HIP graph.

    With ``optimizer`` = the torch.optim.Adam that owns exactly the net's (flattened) parameters, the Adam update is part
    of the sequence too (``fused_adam``): one bridges_adam_step launch over the flat parameter / gradient / moment buffers
    instead of torch's multi-tensor launch; the moments stay the optimiser's own state tensors (re-pointed into flat
    buffers), the step count lives in ``adam_step`` and is written back by ``export_state()`` (before a checkpoint or a
    return to ``optimizer.step()``).  The caller must then NOT call ``optimizer.step()`` after ``launch``."""

    WS_FLOATS = 4 << 20

    def __init__(self, net, batch, use_q, use_sf, optimizer=None, mid_stack=True):
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
        self.x_all, self._prepared, self._n_alloc = None, False, 0
        self.ticket = torch.zeros(4, dtype=torch.int32, device=dev)  # word 0: arrival ticket of the loss kernel (re-armed by it)
        # the gradient tensors the launches write.  With flattened parameters (dqn_ops.FlatParameters) they are views of
        # ONE flat buffer laid out like the parameter buffer; either way they are referenced here as well, so a later
        # zero_grad(set_to_none=True) cannot hand their memory to someone else while a captured graph still writes to it
        flat = getattr(net, "_flat_params", None)
        params = [p for lin in self.linears for p in (lin.weight, lin.bias)]
        self.flat, self.grad_flat = None, None
        if flat is not None and all(self._offset(flat.flat, p) is not None for p in params):
            self.flat, self.grad_flat = flat.flat, torch.zeros_like(flat.flat)
        self._grads = []
        for p in params:
            assert p.dtype == torch.float32 and p.is_contiguous()
            if self.grad_flat is not None:
                off = self._offset(self.flat, p)
                p.grad = self.grad_flat[off:off + p.numel()].view_as(p)
            elif p.grad is None:
                p.grad = torch.zeros_like(p)
            assert p.grad.is_contiguous()
            self._grads.append(p.grad)
        self._setup_mid_stack(mid_stack)
        self.fused_adam, self.optimizer = False, None
        if (optimizer is not None and self.grad_flat is not None and self._adam_applies(optimizer, net)
                and {id(p) for p in params} == {id(p) for p in net.parameters()}):
            self._adopt_adam(optimizer, params)

    def _setup_mid_stack(self, enabled=True):
        """The Linear + ReLU layers between the first and the last as ONE launch each way (bridges_mlp_mid_forward /
        _backward: a workgroup per tile of the stack's last layer computes what the tile depends on itself, no traffic between
        workgroups) where the library has that stack (the reference's 256-128-64-128-256); any other stack keeps a launch
        per layer (bit-identical results: ``mid_stack=False`` lets the tests compare the two)."""
        self.mid = None
        mids = self.linears[1:-1]
        if not mids or not enabled:
            return
        n = len(mids)
        dims = (C.c_int32 * (n + 1))(mids[0].in_features, *[l.out_features for l in mids])
        if not self.L.bridges_mlp_mid_supported(self.rows, n, dims):
            return
        VP, VP1 = C.c_void_p * n, C.c_void_p * (n + 1)
        ptr = lambda t: t.data_ptr()
        self.mid = dict(n=n, dims=dims, W=VP(*[ptr(l.weight) for l in mids]), bias=VP(*[ptr(l.bias) for l in mids]),
                        dW=VP(*[ptr(l.weight.grad) for l in mids]), db=VP(*[ptr(l.bias.grad) for l in mids]),
                        acts=VP1(*[ptr(t) for t in self.acts[1:n + 2]]), dz=VP1(*[ptr(t) for t in self.dz[0:n + 1]]))

    def allocate_inputs(self, n_batches):
        """Room for the first layer's input rows of n_batches batches (52 MB for 25 batches of 32 rows of 64x64 images): after
        this, ``prepare_inputs`` + ``launch`` replace the per-step build of the rows (one launch per optimiser step)."""
        K = self.linears[0].in_features
        self.x_all = torch.zeros((int(n_batches) * self.rows, K), dtype=torch.float32, device=self.linears[0].weight.device)
        self._n_alloc, self._prepared = int(n_batches), False

    def prepare_inputs(self, n_batches, block_all, action_all, binary_all, reward, obstacle):
        """Build the input rows of batches 0 .. n_batches - 1 of the per-call arrays in ONE launch (bridges_mlp_input_batches);
        ``launch`` then reads batch ``counter`` of them."""
        assert self.x_all is not None and n_batches <= self._n_alloc
        for t in (block_all, action_all, binary_all, reward, obstacle):
            assert t.dtype == torch.float32 and t.is_contiguous()
        abi.check(self.L.bridges_mlp_input_batches(int(n_batches), self.batch, self.rows, self.px, self.nf, _ptr(block_all),
                                                   _ptr(action_all), _ptr(binary_all), _ptr(reward), _ptr(obstacle), _ptr(self.x_all),
                                                   _stream()), "bridges_mlp_input_batches")
        self._prepared = True

    @staticmethod
    def _offset(flat, p):
        """Element offset of parameter ``p`` inside the flat buffer, or None if it does not live there."""
        d = p.data_ptr() - flat.data_ptr()
        if d < 0 or d % 4 or d // 4 + p.numel() > flat.numel():
            return None
        return d // 4

    @staticmethod
    def _adam_applies(opt, net):
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        if g.get("amsgrad") or g.get("weight_decay") or g.get("maximize") or g.get("differentiable"):
            return False
        if isinstance(g["lr"], torch.Tensor):
            return False
        return {id(p) for p in g["params"]} == {id(p) for p in net.parameters()}

    def _adopt_adam(self, opt, params):
        g = opt.param_groups[0]
        self.lr, (self.beta1, self.beta2), self.eps = float(g["lr"]), (float(b) for b in g["betas"]), float(g["eps"])
        self.m_flat, self.v_flat = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        step = 0.0
        for p in params:
            off = self._offset(self.flat, p)
            st = opt.state[p]
            mv, vv = self.m_flat[off:off + p.numel()].view_as(p), self.v_flat[off:off + p.numel()].view_as(p)
            if "exp_avg" in st:                                      # continue from what optimizer.step() has done so far
                mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
                step = float(st["step"])
            else:
                st["step"] = torch.zeros((), dtype=torch.float32, device=self.flat.device)
            st["exp_avg"], st["exp_avg_sq"] = mv, vv                 # the optimiser's state IS the flat buffers from here on
        self.adam_step = torch.full((), step, dtype=torch.float32, device=self.flat.device)
        # per layer: the flat range [lo, hi) of its weight + bias (hi rounded up to the 16-byte padding of the layout) and
        # the moment views of the first layer (whose update is folded into its weight-gradient tiles)
        self._slices, self._moments = [], []
        for lin in self.linears:
            ow, ob = self._offset(self.flat, lin.weight), self._offset(self.flat, lin.bias)
            lo, hi = min(ow, ob), max(ow + lin.weight.numel(), ob + lin.bias.numel())
            hi = min((hi + 3) // 4 * 4, self.flat.numel())
            assert lo % 4 == 0
            self._slices.append((lo, hi))
            self._moments.append((opt.state[lin.weight]["exp_avg"], opt.state[lin.weight]["exp_avg_sq"],
                                  opt.state[lin.bias]["exp_avg"], opt.state[lin.bias]["exp_avg_sq"]))
        covered = sorted(self._slices)
        assert all(a[1] <= b[0] for a, b in zip(covered, covered[1:])), "layer ranges of the flat buffer overlap"
        # everything behind the first layer's range: the other layers (and padding, whose gradients and moments stay 0)
        first = self._slices[0]
        self._rest = (first[1], self.flat.numel()) if first == covered[0] and first[0] == 0 and self.flat.numel() % 4 == 0 else None
        # with the middle-layer stack, the head's range (the last of the buffer: 2.1 M of the 2.2 M parameters behind the first
        # layer) is updated by extra workgroups of the stack's backward launch -- its gradient is complete and its weights are
        # not read any more by then --, the last launch keeps the first layer's folded update and the small rest
        self._rest_head = None
        head = self._slices[-1]
        if (self._rest is not None and self.mid is not None and len(self.linears) >= 3 and head == covered[-1]
                and head[0] >= first[1]):
            self._rest_head = (head[0], self.flat.numel())
            self._rest = (first[1], head[0])
        self.optimizer, self.fused_adam = opt, True

    def export_state(self):
        """Write the step count back into the optimiser's per-parameter state (its moments already are the flat buffers):
        call before ``optimizer.state_dict()`` / ``optimizer.step()``."""
        if self.fused_adam:
            for st in self.optimizer.state.values():
                if isinstance(st.get("step"), torch.Tensor):
                    st["step"].copy_(self.adam_step)
                else:
                    st["step"] = float(self.adam_step)

    def check_hyperparameters(self):
        """lr, betas and eps were read off the optimiser when it was adopted and are launch arguments since (a captured graph
        holds them as constants): a later change of the param group would be ignored silently, so it is refused instead."""
        if self.fused_adam:
            g = self.optimizer.param_groups[0]
            now = (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]))
            if now != (self.lr, self.beta1, self.beta2, self.eps):
                raise RuntimeError(f"the optimiser's hyper-parameters changed after the fused step adopted them "
                                   f"({(self.lr, self.beta1, self.beta2, self.eps)} -> {now}): rebuild the step (VecDQN re-captures "
                                   "its graph when _graph_state is reset)")

    def launch(self, counter, block_all, action_all, binary_all, reward, obstacle, q_target_all, sf_target_all, losses):
        L, rows, px, nf, B = self.L, self.rows, self.px, self.nf, self.batch
        st = _stream()
        for t in (block_all, action_all, binary_all, reward, obstacle):
            assert t.dtype == torch.float32 and t.is_contiguous()
        assert counter.dtype == torch.int64 and losses.dtype == torch.float32
        # the first layer's input: batch `counter` of the pre-built rows of all batches (prepare_inputs), else built here
        pre = self.x_all is not None and self._prepared
        x0, blk = (self.x_all, _ptr(counter)) if pre else (self.acts[0], None)
        if not pre:
            abi.check(L.bridges_mlp_input(B, rows, px, nf, _ptr(counter), _ptr(block_all), _ptr(action_all), _ptr(binary_all),
                                          _ptr(reward), _ptr(obstacle), _ptr(self.acts[0]), st), "bridges_mlp_input")
        last = len(self.linears) - 1
        mid = self.mid
        for l, lin in enumerate(self.linears):
            if mid is not None and 0 < l < last:
                if l == 1:
                    abi.check(L.bridges_mlp_mid_forward(rows, mid["n"], mid["dims"], mid["W"], mid["bias"], mid["acts"], st),
                              "bridges_mlp_mid_forward")
                continue
            abi.check(L.bridges_linear_forward(rows, lin.in_features, lin.out_features, _ptr(x0 if l == 0 else self.acts[l]),
                                               _ptr(lin.weight), _ptr(lin.bias), int(l < last), _ptr(self.acts[l + 1]), _ptr(self.ws),
                                               self.ws.numel(), blk if l == 0 else None, st), "bridges_linear_forward")
        # logging the loss, advancing the batch counter and the Adam step count need every row's loss: that rides in the head
        # layer's backward launch (bridges_linear_backward_log, one thread beside its jobs) -- the loss kernel itself hands
        # nothing between workgroups.  (A net whose head is also its first layer keeps the loss kernel's ticket form.)
        log_in_backward = last >= 1
        abi.check(L.bridges_successor_loss(B, rows, px, nf, _ptr(self.acts[-1]), _ptr(reward), _ptr(counter),
                                           _ptr(q_target_all) if self.use_q else None,
                                           _ptr(sf_target_all) if self.use_sf else None, int(self.use_q), int(self.use_sf),
                                           _ptr(self.dz[-1]), _ptr(self.loss_rows), _ptr(self.q),
                                           None if log_in_backward else _ptr(losses), losses.numel(),
                                           None if log_in_backward else _ptr(counter), None if log_in_backward else _ptr(self.ticket),
                                           _ptr(self.adam_step) if (self.fused_adam and not log_in_backward) else None, st),
                  "bridges_successor_loss")
        # the whole optimiser update rides in the LAST backward launch (the first layer's, which has no input gradient):
        # Adam goes into its weight-gradient tiles and extra workgroups of that launch update the other layers, whose
        # gradients are complete and whose weights nothing reads any more.  (Adam launches on a parallel branch of the
        # captured graph were measured: every cross-branch edge costs ~20 us, 245 us per step against 155.)
        fold_first = self.fused_adam and rows == 32 and self._rest is not None
        for l in range(last, -1, -1):
            lin = self.linears[l]
            if mid is not None and 0 < l < last:
                if l == last - 1:
                    rider = self._rest_head if (self.fused_adam and fold_first) else None
                    if rider is not None:
                        lo_h, hi_h = rider
                        offh = lambda t: C.c_void_p(t.data_ptr() + 4 * lo_h)
                        abi.check(L.bridges_mlp_mid_backward(rows, mid["n"], mid["dims"], mid["W"], mid["dW"], mid["db"], mid["acts"],
                                                             mid["dz"], offh(self.flat), offh(self.grad_flat), offh(self.m_flat),
                                                             offh(self.v_flat), hi_h - lo_h, _ptr(self.adam_step), self.lr, self.beta1,
                                                             self.beta2, self.eps, st), "bridges_mlp_mid_backward")
                    else:
                        abi.check(L.bridges_mlp_mid_backward(rows, mid["n"], mid["dims"], mid["W"], mid["dW"], mid["db"], mid["acts"],
                                                             mid["dz"], None, None, None, None, 0, None, 0.0, 0.0, 0.0, 0.0, st),
                                  "bridges_mlp_mid_backward")
                continue
            if l == 0 and fold_first:
                mw, vw, mb, vb = self._moments[0]
                lo, hi = self._rest
                off = lambda t: C.c_void_p(t.data_ptr() + 4 * lo)
                abi.check(L.bridges_linear_backward_adam(rows, lin.in_features, lin.out_features, _ptr(self.dz[0]), _ptr(x0),
                                                         _ptr(lin.weight), _ptr(lin.bias), _ptr(mw), _ptr(vw), _ptr(mb), _ptr(vb),
                                                         off(self.flat), off(self.grad_flat), off(self.m_flat), off(self.v_flat),
                                                         hi - lo, _ptr(self.adam_step), self.lr, self.beta1, self.beta2, self.eps, blk,
                                                         -1, st), "bridges_linear_backward_adam")      # -1: the loss kernel advanced the counter
                continue
            if l == last and log_in_backward:
                abi.check(L.bridges_linear_backward_log(rows, lin.in_features, lin.out_features, _ptr(self.dz[l]), _ptr(self.acts[l]),
                                                        _ptr(lin.weight), _ptr(lin.weight.grad), _ptr(lin.bias.grad), _ptr(self.acts[l]),
                                                        _ptr(self.dz[l - 1]), _ptr(self.ws), self.ws.numel(), _ptr(self.loss_rows), B,
                                                        _ptr(losses), losses.numel(), _ptr(counter),
                                                        _ptr(self.adam_step) if self.fused_adam else None, st),
                          "bridges_linear_backward_log")
                continue
            abi.check(L.bridges_linear_backward(rows, lin.in_features, lin.out_features, _ptr(self.dz[l]),
                                                _ptr(x0 if l == 0 else self.acts[l]), _ptr(lin.weight), _ptr(lin.weight.grad),
                                                _ptr(lin.bias.grad), _ptr(self.acts[l]) if l > 0 else None,
                                                _ptr(self.dz[l - 1]) if l > 0 else None, _ptr(self.ws), self.ws.numel(),
                                                blk if l == 0 else None, -1 if l == 0 else 0, st), "bridges_linear_backward")
        if self.fused_adam and not fold_first:               # larger batches: one flat launch behind the backward pass
            abi.check(L.bridges_adam_step(_ptr(self.flat), _ptr(self.grad_flat), _ptr(self.m_flat), _ptr(self.v_flat),
                                          self.flat.numel(), _ptr(self.adam_step), self.lr, self.beta1, self.beta2, self.eps, st),
                      "bridges_adam_step")
