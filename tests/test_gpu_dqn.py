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


def test_bits_dot_and_accumulate_match_the_dense_count_images():
    """bridges_bits_dot / bridges_bits_accumulate: EpsilonGreedy's overlap  sum(step_images[step] * a)  and its update
    step_images[step] += a_sel  (successor_dqn.py:127-131) on bit-packed rasters, against the dense float images -- exact
    (the count images hold small integers)."""
    from bridges_hip import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(5)
    n_img, n, slots = 60, 500, 11
    dense = (torch.rand((n_img, 64, 64), generator=g) < 0.02)
    dense[3] = False
    dense[4] = True
    weights = (1 << torch.arange(63, dtype=torch.int64))
    bits = (dense[:, :, :63].to(torch.int64) * weights).sum(dim=2)
    bits = torch.where(dense[:, :, 63], bits | torch.tensor(-(1 << 63), dtype=torch.int64), bits)
    img = torch.randint(0, 7, (slots, 64, 64), generator=g).float()
    rows = torch.randint(0, n_img, (n,), generator=g)
    slot = torch.randint(0, slots, (n,), generator=g)
    got = ops.bits_dot(bits.to(dev), img.to(dev), slot.to(dev), bits_row=rows.to(dev))
    want = (img[slot] * dense[rows].float()).sum(dim=(1, 2))
    assert torch.equal(got.cpu(), want)
    w = (torch.rand(n, generator=g) < 0.6).float()
    acc = img.clone().to(dev)
    ops.bits_accumulate_(acc, bits.to(dev), slot.to(dev), weight=w.to(dev), bits_row=rows.to(dev))
    ref = img.clone()
    ref.index_add_(0, slot, dense[rows].float() * w[:, None, None])
    assert torch.equal(acc.cpu(), ref)
    # identity rows, unit weights
    acc2 = torch.zeros((n_img, 64, 64), device=dev)
    ops.bits_accumulate_(acc2, bits.to(dev), torch.arange(n_img, device=dev))
    assert torch.equal(acc2.cpu(), dense.float())


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


@pytest.mark.parametrize("n,N", [(1, 4096), (129, 4096), (1000, 100), (4099, 4096)])
def test_head_sigmoid_dot_matches_the_two_pass_head(n, N):
    """bridges_head_sigmoid_dot: sum_j w[j] * sigmoid(h[r] . Wd[j] + bd[j]) with the product on the f32 matrix cores and never
    stored; tolerance 1e-5 relative to the float64 value (f32 products, f32 accumulation over K = 256, N terms)."""
    from bridges_hip import ops
    dev = torch.device("cuda")
    g = torch.Generator(device="cpu").manual_seed(n + N)
    h = torch.relu(torch.randn((n, 256), generator=g)).to(dev)
    Wd = (torch.randn((N, 256), generator=g) * 0.05).to(dev)
    bd = (torch.randn(N, generator=g) * 0.1).to(dev)
    w = (torch.rand(N, generator=g) * 0.02).to(dev)
    got = ops.head_sigmoid_dot(h, Wd, bd, w)
    want = (torch.sigmoid(h.double() @ Wd.double().T + bd.double()) * w.double()).sum(dim=1)
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-5), float((got.double() - want).abs().max())
    padded = torch.zeros((n, 320), device=dev)
    padded[:, :256] = h                                                         # row stride != K
    assert torch.equal(ops.head_sigmoid_dot(padded[:, :256], Wd, bd, w), got)


@pytest.mark.parametrize("loss_fct", ["mse_q_values", "mse_block_features", "mse_q_values+mse_block_features"])
def test_train_policy_net_reproduces_the_reference_net_fixture(loss_fct, golden_dir):
    """tests/golden/dqn_fixtures.pt (made by make_dqn_fixtures.py from the REFERENCE's own SuccessorMLP / init_weights
    and the restatement of train_policy_net): the same hand-built batch, initial weights and Adam through the product's
    train_policy_net (HIP td-target op) and update_target_net (HIP soft update) -> per-step losses and parameter
    checksums."""
    import os
    import warnings
    from robotoddler.models.cv import SuccessorMLP
    from robotoddler.training.successor_dqn import Transition, flatten_nets, train_policy_net, update_target_net
    warnings.filterwarnings("ignore", message="Using a target size")
    fx = torch.load(os.path.join(golden_dir, "dqn_fixtures.pt"), weights_only=True)
    b, case = fx["batch"], fx["cases"][loss_fct]
    size = tuple(fx["size"])
    mk = lambda: SuccessorMLP(img_size=size, hidden_dims=fx["hidden"])
    pol, tgt = mk().to(DEV), mk().to(DEV)
    pol.load_state_dict(fx["init_state"])
    tgt.load_state_dict(fx["init_state"])
    flatten_nets(pol, tgt)
    opt = torch.optim.Adam(pol.parameters(), lr=fx["lr"])
    na = [int(x) for x in b["num_actions"]]

    class FixedBatch:                                          # replay buffer that always returns the fixture batch
        def __len__(self):
            return 1000

        def sample(self, batch_size=None, stack_tensors=True, device=None):
            d = lambda t: t.to(device)
            return None, Transition(
                block_features=d(b["block"]), binary_features=d(b["binary"]), action=None, action_features=d(b["action"]),
                reward=None, lin_reward=d(b["lin_reward"]), done=tuple(bool(x) for x in b["done"]),
                reward_features=d(b["reward"]), obstacle_features=d(b["obstacle"]), next_block_features=d(b["next_block"]),
                next_binary_features=d(b["next_binary"]), next_available_actions=[[0] * n for n in na],
                next_actions_features=d(b["next_action"]), next_reward_features=d(b["next_reward"]),
                next_obstacle_features=d(b["next_obstacle"]), td_error=None)

    losses = train_policy_net(pol, tgt, opt, FixedBatch(), fx["gamma"], loss_fct=loss_fct, n_steps=3, batch_size=len(na), device=DEV)
    update_target_net(pol, tgt, fx["tau"])
    np.testing.assert_allclose(losses, case["losses"], rtol=1e-5, atol=1e-6)
    for k, v in pol.state_dict().items():
        assert abs(float(v.double().sum()) - case["policy_checksum"][k]) <= 1e-4 * (1.0 + case["policy_abs_checksum"][k]), k
    for k, v in tgt.state_dict().items():
        assert abs(float(v.double().sum()) - case["target_checksum"][k]) <= 1e-4 * (1.0 + case["policy_abs_checksum"][k]), k


def test_epsilon_greedy_follows_the_reference_rule():
    """EpsilonGreedy (successor_dqn.py:98-132): exploit = argmax q, explore = argmin of the overlap with the count image
    of the episode step (first minimum), then the image grows by the chosen raster; epsilon decays per step() call.
    Same random.random() stream for the product class and the restatement, 40 calls over 4 episode steps."""
    from robotoddler.training.successor_dqn import EpsilonGreedy
    g = torch.Generator().manual_seed(5)
    calls = []
    for i in range(40):
        A = int(torch.randint(1, 9, (1,), generator=g))
        feats = (torch.rand(A, 1, 64, 64, generator=g) > 0.9).float()
        if i % 7 == 3 and A > 1:
            feats[1] = feats[0]                                 # a tie: the first minimum must win
        calls.append((torch.randn(A, generator=g), i % 4, feats))
    for eps_start in (1.0, 0.6, 0.0):
        random.seed(11)
        prod = EpsilonGreedy(eps_start=eps_start, eps_end=0.05 if eps_start else 0.0, gamma=0.9, max_steps=4, device=torch.device(DEV))
        got = []
        for k, (q, step, feats) in enumerate(calls):
            got.append(prod(q.to(DEV), step, feats.to(DEV)))
            if k % 5 == 4:
                prod.step()
        random.seed(11)
        ref = O.EpsilonGreedyOracle(eps_start=eps_start, eps_end=0.05 if eps_start else 0.0, gamma=0.9, max_steps=4)
        want = []
        for k, (q, step, feats) in enumerate(calls):
            want.append(ref(q, step, feats))
            if k % 5 == 4:
                ref.step()
        assert got == want, eps_start
        assert prod.epsilon == ref.epsilon
        for a, bb in zip(prod.step_images, ref.step_images):
            assert torch.equal(a.cpu(), bb)
        if eps_start == 1.0:
            assert sum(float(im.sum()) for im in ref.step_images) > 0          # it did explore
        if eps_start == 0.0:
            assert got == [int(torch.argmax(q)) for q, _, _ in calls]


def test_log_episode_reports_the_reference_numbers():
    """log_episode (successor_dqn.py:479-503): discounted reward / lin_reward sums, mean loss, step count, the stable /
    collision flags of the last next state, epsilon -- on hand-made transitions and on a real greedy rollout."""
    from assembly_gym.envs.assembly_env import AssemblyEnv
    from assembly_gym.envs.gym_env import AssemblyGym, bridge_setup, sparse_reward
    from robotoddler.models.cv import SuccessorMLP
    from robotoddler.training.successor_dqn import EpsilonGreedy, Transition, log_episode, rollout_episode
    from robotoddler.utils.utils import init_weights
    blank = dict(block_features=None, binary_features=None, action=None, action_features=None, done=False, reward_features=None,
                 obstacle_features=None, next_block_features=None, next_available_actions=None, next_actions_features=None,
                 next_reward_features=None, next_obstacle_features=None, td_error=0)
    rewards, lins = [-1.0, -1.0, 0.0, 1.0], [0.25, 0.0, 0.0125, 1.5]
    nb = [torch.tensor([[1., 0, 0, 0, 0, 0]]), torch.tensor([[1., 0, 0, 0, 0, 0]]), torch.tensor([[1., 0, 0, 0, 0, 0]]),
          torch.tensor([[0., 1, 0, 0, 0, 0], [0., 1, 0, 0, 0, 0]])]
    trans = [Transition(reward=torch.Tensor([r]), lin_reward=torch.tensor([[l]]), next_binary_features=b, **blank)
             for r, l, b in zip(rewards, lins, nb)]
    pol = EpsilonGreedy(eps_start=0.3, device=torch.device("cpu"))
    info, fig = log_episode(7, trans, [0.5, 0.25, 0.75], 0.8, policy=pol)
    want = O.log_episode_values(rewards, lins, [0.5, 0.25, 0.75], 0.8, nb[-1][0].tolist(), epsilon=pol.epsilon)
    assert fig is None and set(info) == set(want)
    for k in want:
        assert info[k] == pytest.approx(want[k], rel=1e-6), k
    assert log_episode(8, trans[:1], None, 0.8)[0]["avg_loss"] is None
    # a real rollout on the drop-in API
    torch.manual_seed(0)
    random.seed(0)
    net = SuccessorMLP(img_size=(64, 64), hidden_dims=[32, 16, 32]).to(DEV)
    net.apply(init_weights)
    env = AssemblyGym(reward_fct=sparse_reward, max_steps=6, restrict_2d=True, assembly_env=AssemblyEnv(render=False))
    greedy = lambda q, *a, **k: torch.argmax(q)
    transitions, _ = rollout_episode(env=env, policy=greedy, policy_net=net, setup_fct=lambda: bridge_setup(num_stories=2),
                                     x_discr_ground=np.linspace(-2, 0, 10), xlim=(-3, 7), ylim=(0., 10), offset_values=[0],
                                     img_size=(64, 64), device=torch.device(DEV))
    info, _ = log_episode(1, transitions, [1.0, 3.0], 0.8)
    want = O.log_episode_values([float(t.reward) for t in transitions], [float(t.lin_reward) for t in transitions], [1.0, 3.0], 0.8,
                                transitions[-1].next_binary_features[0].tolist())
    assert info["num_steps"] == len(transitions) >= 1
    for k in want:
        assert info[k] == pytest.approx(want[k], rel=1e-5, abs=1e-7), k


@pytest.mark.parametrize("model", ["ConvNet", "Policy"])
def test_fused_conv_epilogues_equal_the_module_forward(model):
    """Inference passes of the conv nets run every convolution bias-free and apply ``+ bias -> ReLU [-> MaxPool2d(2)]``
    as one HIP pass (bridges_bias_relu / bridges_bias_relu_pool2): the ops alone are bit-identical to torch's.  Through
    ConvNet / Policy the 64-wide 16-channel layers additionally run on the hand-written convolution
    (bridges_conv3x3_relu_o16), whose float32 summation order differs from the library's: the outputs agree with the
    plain torch forward (which the same modules take whenever autograd records) to 1e-5 of the largest value."""
    from bridges_hip import dqn_ops
    from robotoddler.models.cv import ConvNet, Policy
    from robotoddler.utils.utils import init_weights
    g = torch.Generator().manual_seed(3)
    for (n, C, H, W) in ((5, 16, 64, 64), (3, 128, 8, 8), (2, 32, 16, 32)):
        x = torch.randn(n, C, H, W, generator=g).to(DEV)
        b = torch.randn(C, generator=g).to(DEV)
        want = torch.relu(x + b[None, :, None, None])
        assert torch.equal(dqn_ops.bias_relu_pool2(x, b), torch.nn.functional.max_pool2d(want, 2))
        assert torch.equal(dqn_ops.bias_relu_(x.clone(), b), want)
    torch.manual_seed(2)
    net = (ConvNet(img_size=(64, 64)) if model == "ConvNet" else Policy()).to(DEV)
    net.apply(init_weights)
    net.eval()
    n = 37
    args = [(torch.rand(n, 1, 64, 64, generator=g) > 0.9).float().to(DEV), (torch.rand(n, 6, generator=g) > 0.5).float().to(DEV),
            (torch.rand(n, 1, 64, 64, generator=g) > 0.95).float().to(DEV), torch.rand(n, 1, 64, 64, generator=g).to(DEV),
            (torch.rand(n, 1, 64, 64, generator=g) > 0.9).float().to(DEV)]
    with torch.no_grad():
        fused = net(*args)
    with torch.enable_grad():
        plain = net(*args)
    for a, bb in zip(fused, plain):
        if a is not None:
            bb = bb.detach()
            assert float((a - bb).abs().max()) <= 1e-5 * max(1.0, float(bb.abs().max())), float((a - bb).abs().max())


@pytest.mark.parametrize("c_in,H,n,pool", [(4, 64, 5, False), (4, 64, 3, True), (16, 64, 4, False), (16, 64, 6, True),
                                           (32, 64, 3, False), (32, 64, 2, True), (16, 8, 3, True), (16, 128, 1, False),
                                           (2, 64, 3, True), (1, 16, 2, False), (3, 64, 2, False)])
def test_hand_written_conv3x3_matches_torch(c_in, H, n, pool):
    """bridges_conv3x3_relu_o16 (f32 matrix cores, LDS-staged bands) against torch's conv2d + relu [+ max_pool2d] in
    float64: same function up to float32 summation order (1e-5 of the largest output), borders included."""
    import torch.nn.functional as F
    from bridges_hip import dqn_ops
    g = torch.Generator(device="cuda").manual_seed(c_in * 100 + H + n)
    x = torch.randn(n, c_in, H, 64, device="cuda", generator=g)
    x[:, :, :, :3] += 1.0                                   # structure at the borders
    w = torch.randn(16, c_in, 3, 3, device="cuda", generator=g) / (3.0 * c_in ** 0.5)
    b = torch.randn(16, device="cuda", generator=g)
    out = dqn_ops.conv3x3_relu_o16(x, w, b, pool)
    ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    if pool:
        ref = F.max_pool2d(ref, 2)
    assert out.shape == ref.shape
    err = float((out.double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err
    assert bool(((out == 0) == (ref.float() <= 0)).float().mean() > 0.999)        # the ReLU pattern agrees


def test_hand_written_conv3x3_unet_variants_match_torch():
    """The U-Net's neighbours folded into the kernel (cv.py:184-197): plain + pooled output from one launch, the
    decoder's torch.cat([up, skip]) read from the two tensors, the 1x1 outconv to one channel as epilogue."""
    import torch.nn.functional as F
    from bridges_hip import dqn_ops
    g = torch.Generator(device="cuda").manual_seed(5)
    n, H = 3, 64
    r = lambda *s: torch.randn(*s, device="cuda", generator=g)
    x, x2 = r(n, 16, H, 64), r(n, 16, H, 64)
    w16, w32, b = r(16, 16, 3, 3) / 12.0, r(16, 32, 3, 3) / 17.0, r(16)
    err = lambda a, ref: float((a.double() - ref).abs().max() / ref.abs().max())
    ref = F.relu(F.conv2d(x.double(), w16.double(), b.double(), padding=1))
    plain, pooled = dqn_ops.conv3x3_relu_o16(x, w16, b, both=True)
    assert err(plain, ref) < 1e-5 and err(pooled, F.max_pool2d(ref, 2)) < 1e-5
    assert torch.equal(pooled, F.max_pool2d(plain, 2))                  # the two outputs of one launch are consistent
    ref2 = F.relu(F.conv2d(torch.cat([x, x2], dim=1).double(), w32.double(), b.double(), padding=1))
    assert err(dqn_ops.conv3x3_relu_o16(x, w32, b, x2=x2), ref2) < 1e-5
    assert err(dqn_ops.conv3x3_relu_o16(torch.cat([x, x2], dim=1), w32, b), ref2) < 1e-5
    w1, b1 = r(1, 16, 1, 1), r(1)
    ref3 = F.conv2d(ref, w1.double(), b1.double())
    out = dqn_ops.conv3x3_relu_o16(x, w16, b, proj=(w1, b1))
    assert out.shape == (n, 1, H, 64) and err(out, ref3) < 1e-5


@pytest.mark.parametrize("c_in,c_out,H,W,n", [(32, 16, 32, 32, 3), (64, 32, 16, 16, 5), (32, 16, 8, 48, 2)])
def test_hand_written_upconv_matches_torch(c_in, c_out, H, W, n):
    """bridges_upconv2x2 (ConvTranspose2d(kernel 2, stride 2) + bias, cv.py:176, 179) against torch in float64."""
    import torch.nn.functional as F
    from bridges_hip import dqn_ops
    g = torch.Generator(device="cuda").manual_seed(c_in + H)
    x = torch.randn(n, c_in, H, W, device="cuda", generator=g)
    w = torch.randn(c_in, c_out, 2, 2, device="cuda", generator=g) / c_in ** 0.5
    b = torch.randn(c_out, device="cuda", generator=g)
    out = dqn_ops.upconv2x2(x, w, b)
    ref = F.conv_transpose2d(x.double(), w.double(), b.double(), stride=2)
    assert out.shape == ref.shape
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 1e-5


@pytest.mark.parametrize("E,amax,eps,greedy", [(1, 5, 0.5, False), (300, 40, 0.3, False), (300, 40, 1.0, False), (257, 150, 0.3, True)])
def test_eps_greedy_select_matches_the_segmented_torch_formulation(E, amax, eps, greedy):
    """bridges_eps_greedy_select against the per-env rule written out with torch: first maximum of q, or -- when the env's draw
    is <= eps -- first minimum of the overlap; ties (quantised values), empty segments and segments longer than a wave."""
    from bridges_hip import ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(E * 7 + amax)
    counts = torch.randint(0, amax + 1, (E,), generator=g)
    counts[::5] = 0
    if int(counts.sum()) == 0:
        counts[0] = 3
    seg = torch.zeros(E + 1, dtype=torch.int32)
    seg[1:] = torch.cumsum(counts, 0)
    n = int(seg[E])
    q = torch.randint(-3, 4, (n,), generator=g).float() * 0.25              # many ties
    join = torch.randint(0, 5, (n,), generator=g).float()
    u = torch.rand(E, generator=g)
    idx = torch.sort(torch.randperm(n * 3, generator=g)[:n]).values          # compact candidate indices, ascending like the env's
    cand_offset = torch.zeros(E, dtype=torch.int32)
    for e in range(E):
        cand_offset[e] = int(idx[seg[e]]) - (e % 3) if counts[e] else int(idx[min(int(seg[e]), n - 1)]) + 5
    got = ops.eps_greedy_select(seg.to(dev), q.to(dev), join.to(dev), u.to(dev), eps, greedy, idx.to(dev), cand_offset.to(dev))
    sel_compact, sel_index, q_sel, ex_w = [t.cpu() for t in got]
    for e in range(E):
        lo, hi = int(seg[e]), int(seg[e + 1])
        explore = (not greedy) and bool(u[e] <= torch.tensor(eps, dtype=torch.float32))
        if hi > lo:
            row = lo + (int(torch.argmin(join[lo:hi])) if explore else int(torch.argmax(q[lo:hi])))   # torch returns the first extremum
            v = join[lo:hi] if explore else q[lo:hi]
            assert row == lo + int((v == (v.min() if explore else v.max())).nonzero()[0])
        else:
            row = 0
        assert int(sel_compact[e]) == int(idx[row]), e
        assert int(sel_index[e]) == max(int(idx[row]) - int(cand_offset[e]), 0)
        assert float(q_sel[e]) == (float(q[row]) if hi > lo else 0.0)
        assert float(ex_w[e]) == (1.0 if (explore and hi > lo) else 0.0)


@pytest.mark.parametrize("loss_fct,B,size,hidden", [("mse_q_values", 32, (64, 64), [256, 128, 64, 128, 256]),
                                                   ("mse_block_features", 32, (64, 64), [256, 128, 64, 128, 256]),
                                                   ("mse_q_values+mse_block_features", 8, (64, 64), [128, 64, 128])])
def test_train_policy_net_on_the_hand_written_step_follows_the_autograd_form(loss_fct, B, size, hidden):
    """The single-environment train_policy_net (successor_dqn.py:157-277) takes the hand-written SuccessorMLP step when every
    sampled transition carries the same task fingerprint (rollout_episode tags them): per-step losses -- incl. the [B,B]
    broadcast of the q target, i.e. the var(lin_reward) term -- within 1e-5 of the autograd form on the same batches, the
    weights after three Adam steps within what two correct float32 Adam runs agree to; untagged transitions keep the
    autograd form; the optimiser's state is the true one after sync_fused_optimizer."""
    import warnings
    from robotoddler.models.cv import SuccessorMLP
    from robotoddler.training import successor_dqn as S
    from robotoddler.utils.replay_memory import ReplayBuffer
    from robotoddler.utils.utils import init_weights
    warnings.filterwarnings("ignore", message="Using a target size")
    gamma = 0.8
    trans = synthetic_transitions(3 * B, size, 5)
    reward, obstacle = trans[0].reward_features.to(DEV), trans[0].obstacle_features.to(DEV)
    key = S._task_key(reward, obstacle)
    dev_t = lambda t: t.to(DEV) if torch.is_tensor(t) else t
    shared, plain = [], []
    for t in trans:
        n1 = t.next_block_features.shape[0]
        d = {f: dev_t(getattr(t, f)) for f in t._fields}
        d.update(next_reward_features=reward.expand(n1, -1, -1, -1), next_obstacle_features=obstacle.expand(n1, -1, -1, -1))
        if B == 32:             # the next state's rows as rollout_episode stores them: expand()ed views of one row (gathered by
            d.update(next_block_features=d["next_block_features"][:1].expand(n1, -1, -1, -1),        # run_all; else: concatenated)
                     next_binary_features=d["next_binary_features"][:1].expand(n1, -1))
        plain.append(Transition(**dict(d, reward_features=reward.clone(), obstacle_features=obstacle.clone())))
        shared.append(Transition(**dict(d, reward_features=S._tagged(reward.clone(), key), obstacle_features=S._tagged(obstacle.clone(), key))))
    torch.manual_seed(1)
    mk = lambda: SuccessorMLP(img_size=size, hidden_dims=hidden).to(DEV)
    nets = {}
    for name, items in (("fused", shared), ("autograd", plain)):
        pol, tgt = mk(), mk()
        if nets:
            pol.load_state_dict(nets["fused"][4])
        else:
            pol.apply(init_weights)
        init = {k: v.clone() for k, v in pol.state_dict().items()}
        tgt.load_state_dict(init)
        S.flatten_nets(pol, tgt)
        opt = torch.optim.Adam(pol.parameters(), lr=1e-3)
        rb = ReplayBuffer(capacity=1000)
        rb.push(items)
        random.seed(21)
        losses = S.train_policy_net(pol, tgt, opt, rb, gamma, loss_fct=loss_fct, n_steps=3, batch_size=B, device=DEV)
        nets[name] = (pol, opt, losses, getattr(pol, "_fused_trainer", None), init)
    assert nets["fused"][3] is not None and nets["autograd"][3] is None          # the tagged run took the hand-written step
    np.testing.assert_allclose(nets["fused"][2], nets["autograd"][2], rtol=2e-5, atol=1e-6)
    for pa, pb in zip(nets["autograd"][0].parameters(), nets["fused"][0].parameters()):
        err = float((pa.detach().double() - pb.detach().double()).norm() / (pa.detach().double().norm() + 1e-30))
        assert err < 2e-4, err
    S.sync_fused_optimizer(nets["fused"][0])
    steps = {float(st["step"]) for st in nets["fused"][1].state.values()}
    assert steps == {3.0}
    # a further step by the autograd form on the same net (untagged batch): the optimiser carries on from step 3
    rb = ReplayBuffer(capacity=1000)
    rb.push(plain)
    random.seed(22)
    more = S.train_policy_net(nets["fused"][0], mk(), nets["fused"][1], rb, gamma, loss_fct=loss_fct, n_steps=1, batch_size=B, device=DEV)
    assert len(more) == 1 and np.isfinite(more[0]) and {float(st["step"]) for st in nets["fused"][1].state.values()} == {4.0}


def test_train_policy_net_graph_of_all_steps_equals_the_queued_launches(monkeypatch):
    """From the second call with the same number of steps on, train_policy_net replays its optimiser steps from one HIP graph
    (_FusedTrainer._steps): the same launches on copies of the same inputs -- losses and weights bit-identical to the run that
    queues them one by one (BRIDGES_TRAIN_GRAPH=0), over three calls of four steps."""
    import warnings
    from robotoddler.models.cv import SuccessorMLP
    from robotoddler.training import successor_dqn as S
    from robotoddler.utils.replay_memory import ReplayBuffer
    from robotoddler.utils.utils import init_weights
    warnings.filterwarnings("ignore", message="Using a target size")
    B, size, gamma = 16, (64, 64), 0.9
    trans = synthetic_transitions(4 * B, size, 9)
    reward, obstacle = trans[0].reward_features.to(DEV), trans[0].obstacle_features.to(DEV)
    key = S._task_key(reward, obstacle)
    items = []
    for t in trans:
        n1 = t.next_block_features.shape[0]
        d = {f: (getattr(t, f).to(DEV) if torch.is_tensor(getattr(t, f)) else getattr(t, f)) for f in t._fields}
        d.update(next_reward_features=reward.expand(n1, -1, -1, -1), next_obstacle_features=obstacle.expand(n1, -1, -1, -1),
                 next_block_features=d["next_block_features"][:1].expand(n1, -1, -1, -1),
                 next_binary_features=d["next_binary_features"][:1].expand(n1, -1),
                 reward_features=S._tagged(reward.clone(), key), obstacle_features=S._tagged(obstacle.clone(), key))
        items.append(Transition(**d))
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BRIDGES_TRAIN_GRAPH", mode)
        torch.manual_seed(5)
        pol = SuccessorMLP(img_size=size, hidden_dims=[128, 64, 128]).to(DEV)
        pol.apply(init_weights)
        tgt = SuccessorMLP(img_size=size, hidden_dims=[128, 64, 128]).to(DEV)
        tgt.load_state_dict(pol.state_dict())
        S.flatten_nets(pol, tgt)
        opt = torch.optim.Adam(pol.parameters(), lr=1e-3)
        rb = ReplayBuffer(capacity=1000)
        rb.push(items)
        random.seed(3)
        losses = []
        for _ in range(3):
            losses += S.train_policy_net(pol, tgt, opt, rb, gamma, loss_fct="mse_q_values+mse_block_features", n_steps=4, batch_size=B,
                                         device=DEV)
            S.update_target_net(pol, tgt, 0.05)
        tr = pol._fused_trainer
        assert (4 in tr._graphs) == (mode == "1")
        out[mode] = (losses, torch.cat([p.detach().flatten() for p in pol.parameters()]).cpu())
    assert len(out["1"][0]) == 12 and out["1"][0] == out["0"][0]
    assert torch.equal(out["1"][1], out["0"][1])
