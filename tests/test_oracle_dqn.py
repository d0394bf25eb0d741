"""Cross-checks the float64 DQN oracle (oracle/dqn_oracle.py) against torch-CPU autograd.  CPU only.
(The keras-rl fork / Keras / TF are not importable: the DQN half is 'parity unpinned', see oracle/__init__.py.)"""
import numpy as np
import pytest
import torch

from oracle import dqn_oracle as O

SPECS = {
    "c1": ((4, 7, 7), 10),
    "c3": ((7, 11, 11), 51),
    "c5": ((9, 15, 15), 99),
}
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]


def torch_forward(spec, flat, obs, keep=None):
    """Same network in torch float64 (weights converted HWIO -> OIHW)."""
    P = spec.split(flat)
    x = obs
    mi = 0
    for (kind, L), (Wk, bk) in zip(spec.layers, P):
        if kind == "conv":
            x = torch.relu(torch.nn.functional.conv2d(x, Wk.permute(3, 2, 0, 1), bk, stride=L["s"]))
        else:
            if x.dim() == 4:
                x = x.flatten(1)
            x = x @ Wk + bk
            if L["relu"]:
                x = torch.relu(x)
            if L["dropout"] > 0 and keep is not None:
                x = torch.where(keep[mi], x / (1.0 - L["dropout"]), torch.zeros_like(x))
                mi += 1
    if spec.dueling:
        x = x[:, 0:1] + x[:, 1:] - x[:, 1:].mean(dim=1, keepdim=True)
    return x


@pytest.mark.parametrize("name", ["c1", "c3", "c5"])
def test_param_counts_match_survey(name):
    shape, A = SPECS[name]
    spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A)
    assert spec.n_params == {"c1": 36867, "c3": 193283, "c5": 488499}[name]
    assert spec.forward_macs() == {"c1": 79214, "c3": 444956, "c5": 1121516}[name]


@pytest.mark.parametrize("name,training", [("c1", False), ("c3", True), ("c5", True)])
def test_forward_backward_vs_torch_autograd(name, training):
    shape, A = SPECS[name]
    spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A)
    rng = np.random.RandomState(3)
    flat = O.glorot_init(spec, (1, 2)).astype(np.float64)
    flat += rng.randn(flat.size) * 0.01                    # non-zero biases too
    B = 5
    obs = (rng.rand(B, *shape) < 0.3).astype(np.float64)
    keep = [O.dropout_keep_mask((1, 2), 7, np.arange(B), 512, 0.2)] if training else None
    q, cache = O.forward(spec, flat, obs, training=training, keep_masks=keep)
    tp = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
    tq = torch_forward(spec, tp, torch.tensor(obs), None if keep is None else [torch.tensor(k) for k in keep])
    assert np.allclose(q, tq.detach().numpy(), atol=1e-12)
    dq = rng.randn(B, A)
    g = O.backward(spec, flat, cache, dq)
    (tq * torch.tensor(dq)).sum().backward()
    assert np.allclose(g, tp.grad.numpy(), atol=1e-11)


def test_td_loss_adam_vs_torch():
    rng = np.random.RandomState(0)
    B, A = 16, 51
    q1o, q1t, q0 = rng.randn(B, A), rng.randn(B, A), rng.randn(B, A)
    r, term, act = rng.rand(B) < 0.3, rng.rand(B) < 0.2, rng.randint(0, A, size=B)
    y = O.td_targets(q1o, q1t, r, term, 0.99)
    for b in range(B):
        assert y[b] == r[b] + 0.99 * q1t[b, q1o[b].argmax()] * (0.0 if term[b] else 1.0)
    loss, mean_q, dq = O.loss_and_grad(q0, act, y)
    tq = torch.tensor(q0, requires_grad=True)
    tl = (0.5 * (tq[torch.arange(B), torch.tensor(act)] - torch.tensor(y)) ** 2).mean()
    tl.backward()
    assert np.isclose(loss, tl.item()) and np.allclose(dq, tq.grad.numpy()) and np.isclose(mean_q, q0.max(axis=1).mean())
    # Keras Adam == torch Adam up to where epsilon sits: compare with a hand-rolled torch version
    p, g = rng.randn(100), rng.randn(100)
    m = v = np.zeros(100)
    for t in range(1, 4):
        p2, m, v = O.adam_step(p, g, m, v, t, 1e-3)
        lr_t = 1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        assert np.allclose(p2, p - lr_t * m / (np.sqrt(v) + 1e-7))
        p = p2


def test_policy_and_schedule():
    assert O.annealed_eps(0, 1.0, 0.02, 100000) == 1.0
    assert O.annealed_eps(50000, 1.0, 0.02, 100000) == pytest.approx(0.51)
    assert O.annealed_eps(10 ** 7, 1.0, 0.02, 100000) == 0.02
    q = np.array([0.1, 0.9, 0.9, -1.0])
    assert O.select_action(q, 0b1101, 0.0, False, (0, 2 ** 32 - 1, 0, 0)) == 1       # first maximum, all actions
    assert O.select_action(q, 0b1101, 0.0, True, (0, 2 ** 32 - 1, 0, 0)) == 2        # legal only
    assert O.select_action(q, 0b1101, 1.0, True, (2 ** 31, 0, 0, 0)) == 2            # explore: k = 1 of {0,2,3}
    keep = O.dropout_keep_mask((5, 6), 3, np.arange(64), 512, 0.2)
    assert 0.75 < keep.mean() < 0.85
    assert (O.dropout_keep_mask((5, 6), 3, np.arange(64), 512, 0.2) == keep).all()
    assert (O.dropout_keep_mask((5, 6), 4, np.arange(64), 512, 0.2) != keep).any()
