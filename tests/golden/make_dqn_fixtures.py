"""Generates tests/golden/dqn_fixtures.pt: a hand-built batch pushed through the REFERENCE's importable nets
(robotoddler.models.cv of /root/reference) and the line-by-line restatement of train_policy_net / update_target_net
(oracle/dqn.py; the reference's successor_dqn.py itself needs aim / compas to import): inputs, initial weights, the
loss of each of three Adam steps and checksums of the parameters afterwards, for the three loss functions.
Run in the build container:  python tests/golden/make_dqn_fixtures.py
Tensors and plain numbers only; loaded with torch.load(weights_only=True)."""
import os
import sys
import warnings

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dqn as O                                                     # noqa: E402

sys.path.insert(0, "/root/reference")
from robotoddler.models.cv import SuccessorMLP                                  # noqa: E402  (reference code)
from robotoddler.utils.utils import init_weights                                # noqa: E402  (reference code)

warnings.filterwarnings("ignore", message="Using a target size")
size, B, gamma, tau, lr = (16, 16), 6, 0.8, 0.01, 1e-3
g = torch.Generator().manual_seed(77)
img = lambda *s: (torch.rand(*s, generator=g) > 0.75).float()
num_actions = [3, 1, 4, 2, 1, 5]                          # rows of the next state per transition (max(1, A'))
R = sum(num_actions)
batch = dict(
    block=img(B, 1, *size), binary=(torch.rand(B, 6, generator=g) > 0.5).float(), action=img(B, 1, *size),
    reward=torch.rand(B, 1, *size, generator=g) * 0.05, obstacle=img(B, 1, *size),
    lin_reward=torch.rand(B, 1, generator=g), done=[False, True, False, False, True, False],
    next_block=img(R, 1, *size), next_binary=(torch.rand(R, 6, generator=g) > 0.5).float(), next_action=img(R, 1, *size),
    next_reward=torch.rand(R, 1, *size, generator=g) * 0.05, next_obstacle=img(R, 1, *size), num_actions=num_actions)
out = dict(batch={k: (v if torch.is_tensor(v) else torch.tensor(v)) for k, v in batch.items()}, gamma=gamma, tau=tau, lr=lr,
           size=list(size), hidden=[32, 16, 32], cases={})

for loss_fct in ("mse_q_values", "mse_block_features", "mse_q_values+mse_block_features"):
    torch.manual_seed(5)
    pol, tgt = SuccessorMLP(img_size=size, hidden_dims=[32, 16, 32]), SuccessorMLP(img_size=size, hidden_dims=[32, 16, 32])
    pol.apply(init_weights)
    tgt.load_state_dict(pol.state_dict())
    init = {k: v.clone() for k, v in pol.state_dict().items()}
    opt = torch.optim.Adam(pol.parameters(), lr=lr)
    losses = []
    for _ in range(3):
        q, sf, _ = pol(batch["block"], batch["binary"], batch["action"], batch["reward"], batch["obstacle"])
        with torch.no_grad():
            nq, nsf, _ = tgt(batch["next_block"], batch["next_binary"], batch["next_action"], batch["next_reward"],
                             batch["next_obstacle"])
            q_t, _, st_t, _ = O.td_targets(nq, nsf, num_actions, batch["done"], gamma, batch["lin_reward"], batch["action"])
        loss = O.losses(q, sf, q_t, st_t, loss_fct)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
    new_tgt = O.update_target_net(pol.state_dict(), tgt.state_dict(), tau)
    out["cases"][loss_fct] = dict(
        losses=losses,
        policy_checksum={k: float(v.double().sum()) for k, v in pol.state_dict().items()},
        policy_abs_checksum={k: float(v.double().abs().sum()) for k, v in pol.state_dict().items()},
        target_checksum={k: float(v.double().sum()) for k, v in new_tgt.items()})
    out["init_state"] = init                              # same seed -> same initial weights for the three cases
torch.save(out, os.path.join(HERE, "dqn_fixtures.pt"))
print({k: v["losses"] for k, v in out["cases"].items()})
