"""GPU: fused DQN ops and the training functions against the CPU restatement (oracle/dqn.py) of
robotoddler/training/successor_dqn.py:157-288.  Index outputs bit-exact, float targets/losses to 1e-5."""
import random
from collections import namedtuple

import numpy as np
import pytest
import torch

from oracle import dqn as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def ragged_case(seed, B=7, D=64, with_ties=True):
    g = torch.Generator().manual_seed(seed)
    num_actions = [int(x) for x in torch.randint(1, 9, (B,), generator=g)]
    R = sum(num_actions)
    next_q = torch.randn(R, generator=g)
    if with_ties:
        next_q[1] = next_q[0]                               # first maximum must win
        next_q = torch.round(next_q * 4) / 4
    next_sf = torch.randn(R, 2, 8, D // 8, generator=g)
    done = [bool(x) for x in torch.rand(B, generator=g) < 0.3]
    lin = torch.rand(B, generator=g)
    act = (torch.rand(B, 1, 8, D // 8, generator=g) > 0.5).float()
    return num_actions, next_q, next_sf, done, lin, act


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_td_target_matches_restatement(seed):
    from bridges_hip import dqn_ops
    num_actions, next_q, next_sf, done, lin, act = ragged_case(seed)
    gamma = 0.8
    q_ref, nq_ref, st_ref, sel_ref = O.td_targets(next_q, next_sf, num_actions, done, gamma, lin, act)
    seg = torch.tensor(np.cumsum([0] + num_actions), dtype=torch.int32, device=DEV)
    nsf_dev = next_sf.to(DEV)
    q_t, sf_t, rows = dqn_ops.td_target(seg, next_q.to(DEV), lin.to(DEV), torch.tensor(done, device=DEV), gamma,
                                        next_sf=nsf_dev[:, 0], action_raster=act.squeeze(1).to(DEV))
    assert rows.cpu().tolist() == sel_ref
    torch.testing.assert_close(q_t.cpu(), q_ref, rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(sf_t.cpu().view_as(st_ref), st_ref, rtol=1e-6, atol=1e-6)
    # gamma = 1, lin = 0: the masked next q itself
    z = torch.zeros(len(num_actions), device=DEV)
    q_sel, none, _ = dqn_ops.td_target(seg, next_q.to(DEV), z, torch.tensor(done, device=DEV), 1.0)
    assert none is None
    assert torch.equal(q_sel.cpu(), nq_ref)


def test_td_target_empty_segment_is_masked_when_done():
    from bridges_hip import dqn_ops
    seg = torch.tensor([0, 2, 2, 3], dtype=torch.int32, device=DEV)        # transition 1 has no next action
    nq = torch.tensor([1.0, 3.0, -2.0], device=DEV)
    q, _, _ = dqn_ops.td_target(seg, nq, torch.tensor([0.5, 0.25, 0.0], device=DEV),
                                torch.tensor([False, True, False], device=DEV), 0.5)
    assert q.cpu().tolist() == [0.5 + 0.5 * 3.0, 0.25, -1.0]


@pytest.mark.parametrize("n", [5, 1024, 6387020])
def test_soft_update_bitwise(n):
    from bridges_hip import dqn_ops
    g = torch.Generator().manual_seed(n)
    p, t = torch.randn(n, generator=g), torch.randn(n, generator=g)
    ref = O.update_target_net(dict(w=p), dict(w=t), 0.01)["w"]
    td = t.to(DEV)
    dqn_ops.soft_update_(td, p.to(DEV), 0.01)
    assert torch.equal(td.cpu(), ref)


Transition = namedtuple('Transition', ('block_features', 'binary_features', 'action', 'action_features', 'reward',
                                       'lin_reward', 'done', 'reward_features', 'obstacle_features',
                                       'next_block_features', 'next_binary_features', 'next_available_actions',
                                       'next_actions_features', 'next_reward_features', 'next_obstacle_features',
                                       'td_error'))


def synthetic_transitions(n, size, seed):
    g = torch.Generator().manual_seed(seed)
    img = lambda *s: (torch.rand(*s, generator=g) > 0.7).float()
    out = []
    for k in range(n):
        A = int(torch.randint(0, 5, (1,), generator=g))
        n1 = max(1, A)
        out.append(Transition(
            block_features=img(1, 1, *size), binary_features=img(1, 6), action=None, action_features=img(1, 1, *size),
            reward=torch.Tensor([-1.0]), lin_reward=torch.rand(1, 1, generator=g),       # [1,1] as in rollout_episode
            done=bool(A == 0 or k % 3 == 0), reward_features=torch.rand(1, 1, *size, generator=g),
            obstacle_features=img(1, 1, *size), next_block_features=img(1, 1, *size).expand(n1, -1, -1, -1),
            next_binary_features=img(1, 6).expand(n1, -1), next_available_actions=[None] * A,
            next_actions_features=img(n1, 1, *size), next_reward_features=torch.rand(1, 1, *size, generator=g).expand(n1, -1, -1, -1),
            next_obstacle_features=img(1, 1, *size).expand(n1, -1, -1, -1), td_error=0.0))
    return out


@pytest.mark.parametrize("loss_fct", ["mse_q_values", "mse_block_features", "mse_q_values+mse_block_features"])
def test_train_policy_net_and_update_target_follow_the_reference(loss_fct):
    """3 optimisation steps of train_policy_net + update_target_net on the GPU vs the restatement on the CPU with the
    same nets, batches (random.seed) and Adam: per-step losses to 1e-5 -- including the reference's [B,B] broadcast
    of the q target."""
    import warnings
    from robotoddler.models.cv import SuccessorMLP
    from robotoddler.training.successor_dqn import flatten_nets, train_policy_net, update_target_net
    from robotoddler.utils.replay_memory import ReplayBuffer
    from robotoddler.utils.utils import init_weights
    warnings.filterwarnings("ignore", message="Using a target size")
    size, B, gamma, tau = (16, 16), 6, 0.8, 0.01
    trans = synthetic_transitions(12, size, 3)
    torch.manual_seed(0)
    mk = lambda: SuccessorMLP(img_size=size, hidden_dims=[32, 16, 32])
    pol_c, tgt_c = mk(), mk()
    pol_c.apply(init_weights)
    tgt_c.load_state_dict(pol_c.state_dict())
    pol_g, tgt_g = mk().to(DEV), mk().to(DEV)
    pol_g.load_state_dict(pol_c.state_dict())
    tgt_g.load_state_dict(tgt_c.state_dict())
    flatten_nets(pol_g, tgt_g)
    opt_c = torch.optim.Adam(pol_c.parameters(), lr=1e-3)
    opt_g = torch.optim.Adam(pol_g.parameters(), lr=1e-3)
    rb = ReplayBuffer(capacity=100)
    rb.push(trans)

    random.seed(9)
    losses_g = train_policy_net(pol_g, tgt_g, opt_g, rb, gamma, loss_fct=loss_fct, n_steps=3, batch_size=B, device=DEV)
    update_target_net(pol_g, tgt_g, tau)

    random.seed(9)
    losses_c = []
    for _ in range(3):
        _, batch = rb.sample(batch_size=B, stack_tensors=True, device="cpu")
        q, sf, sb = pol_c(batch.block_features, batch.binary_features, batch.action_features, batch.reward_features,
                          batch.obstacle_features)
        with torch.no_grad():
            nq, nsf, _ = tgt_c(batch.next_block_features, batch.next_binary_features, batch.next_actions_features,
                               batch.next_reward_features, batch.next_obstacle_features)
            num_actions = [max(1, len(a)) for a in batch.next_available_actions]
            q_t, _, st_t, _ = O.td_targets(nq, nsf, num_actions, batch.done, gamma, batch.lin_reward, batch.action_features)
        loss = O.losses(q, sf, q_t, st_t, loss_fct)
        opt_c.zero_grad()
        loss.backward()
        opt_c.step()
        losses_c.append(loss.item())
    new_tgt = O.update_target_net(pol_c.state_dict(), tgt_c.state_dict(), tau)
    np.testing.assert_allclose(losses_g, losses_c, rtol=1e-5, atol=1e-6)
    for k, v in tgt_g.state_dict().items():
        torch.testing.assert_close(v.cpu(), new_tgt[k], rtol=1e-4, atol=1e-5)


def test_update_target_net_per_tensor_path_equals_flat_path():
    from robotoddler.models.cv import Policy
    from robotoddler.training.successor_dqn import flatten_nets, update_target_net
    torch.manual_seed(1)
    a, b = Policy().to(DEV), Policy().to(DEV)
    a2, b2 = Policy().to(DEV), Policy().to(DEV)
    a2.load_state_dict(a.state_dict())
    b2.load_state_dict(b.state_dict())
    ref = O.update_target_net({k: v.cpu() for k, v in a.state_dict().items()}, {k: v.cpu() for k, v in b.state_dict().items()}, 0.05)
    update_target_net(a, b, 0.05)                       # per-tensor launches
    flatten_nets(a2, b2)
    update_target_net(a2, b2, 0.05)                     # one launch
    for k in ref:
        assert torch.equal(b.state_dict()[k].cpu(), ref[k]), k
        assert torch.equal(b2.state_dict()[k].cpu(), ref[k]), k


@pytest.mark.parametrize("d", [16, 256, 516])
def test_bits_linear_matches_the_dense_product(d):
    """bridges_bits_linear: a linear layer over flattened binary rasters fed with the bit-packed rasters equals
    expand-to-f32 @ W (f32 summation order aside), with per-row base rows and an index into the raster array."""
    from bridges_hip import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(d)
    n_img, n, nb = 50, 333, 7
    dense = (torch.rand((n_img, 64, 64), generator=g) < 0.02)
    dense[3] = False                                                  # an empty raster
    dense[4] = True                                                   # a full one
    weights = (1 << torch.arange(63, dtype=torch.int64))
    bits = (dense[:, :, :63].to(torch.int64) * weights).sum(dim=2)
    bits = torch.where(dense[:, :, 63], bits | torch.tensor(-(1 << 63), dtype=torch.int64), bits)      # bit 63 = sign bit
    wt = torch.randn((4096, d), generator=g)
    base = torch.randn((nb, d), generator=g)
    bits_row = torch.randint(0, n_img, (n,), generator=g)
    base_row = torch.randint(0, nb, (n,), generator=g)
    got = ops.bits_linear(bits.to(dev), wt.to(dev), bits_row=bits_row.to(dev), base=base.to(dev), base_row=base_row.to(dev))
    want = dense[bits_row].reshape(n, 4096).double() @ wt.double() + base[base_row].double()
    assert got.shape == (n, d)
    assert torch.allclose(got.cpu().double(), want, rtol=1e-5, atol=1e-4), float((got.cpu().double() - want).abs().max())
    # integer-valued weights (the exploration count images): exact
    cnt = torch.randint(0, 5, (4096, 16), generator=g).float()
    got = ops.bits_linear(bits.to(dev), cnt.to(dev))
    assert torch.equal(got.cpu(), dense.reshape(n_img, 4096).float() @ cnt)


def test_sigmoid_dot_matches_torch():
    """bridges_sigmoid_dot: sum_j w[j] * sigmoid(d[r, j]) in one pass (the q head of the factored SuccessorMLP forward)."""
    from bridges_hip import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(0)
    d = (torch.randn((777, 4096), generator=g) * 3).to(dev)
    w = (torch.rand(4096, generator=g) * 0.02).to(dev)
    got = ops.sigmoid_dot(d, w)
    want = (torch.sigmoid(d.double()) * w.double()).sum(dim=1)
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-5), float((got.double() - want).abs().max())
    view = torch.randn((50, 8192), generator=g).to(dev)[:, 4096:]             # row stride != k
    assert torch.allclose(ops.sigmoid_dot(view, w).double(), (torch.sigmoid(view.double()) * w.double()).sum(dim=1), rtol=1e-5, atol=1e-5)
