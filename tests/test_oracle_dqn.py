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
        m = x[:, 1:].mean() if getattr(spec, "dueling_mean", "row") == "batch" else x[:, 1:].mean(dim=1, keepdim=True)
        x = x[:, 0:1] + x[:, 1:] - m
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


# ---- keras-rl replay memory restatement (oracle/memory_oracle.py) -------------------------------------------------------------------
def test_memory_oracle_ring_and_sampling_rule():
    """RingBuffer drops the oldest entry; sample() returns experiences (entry idx-1 -> entry idx) with idx in [2, nb_entries-1], never
    one whose predecessor entry is terminal; state1 is the next stored observation; terminal1 is entry idx-1's flag."""
    import random
    import warnings
    from oracle import memory_oracle as M
    rb = M.RingBuffer(4)
    for i in range(7):
        rb.append(i)
    assert len(rb) == 4 and [rb[i] for i in range(4)] == [3, 4, 5, 6]
    with pytest.raises(KeyError):
        rb[4]
    rng = random.Random(3)
    nrng = np.random.RandomState(0)
    mem = M.SequentialMemory(limit=40, window_length=1)
    flags = []
    for k in range(57):                                         # wraps: entries 17..56 survive
        term = bool(nrng.rand() < 0.2)
        mem.append(k, k % 5, float(k), term)
        flags.append(term)
    assert mem.nb_entries == 40 and mem.observations[0] == 17
    kept = flags[17:]
    expect = [i for i in range(2, 40) if not kept[i - 2]]
    assert mem.valid_idxs() == expect and 1 not in expect and 39 in expect or kept[37]
    seen = set()
    for _ in range(300):
        for e in mem.sample(8, rng=rng):
            assert e["idx"] in expect
            assert e["state0"] == [17 + e["idx"] - 1] and e["state1"] == [17 + e["idx"]]
            assert e["action"] == (17 + e["idx"] - 1) % 5 and e["reward"] == float(17 + e["idx"] - 1)
            assert e["terminal1"] == kept[e["idx"] - 1]
            seen.add(e["idx"])
    assert seen == set(expect)                                   # the support is exactly the valid set
    # first draws are distinct when the range allows it (random.sample), with replacement + warning otherwise
    idxs = M.sample_batch_indexes(1, 39, 38, rng=rng)
    assert sorted(idxs) == list(range(1, 39))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        idxs = M.sample_batch_indexes(1, 4, 10, rng=rng, np_rng=nrng)
        assert len(w) == 1 and len(idxs) == 10 and set(idxs) <= {1, 2, 3}
    small = M.SequentialMemory(limit=10)
    small.append(0, 0, 0.0, False)
    small.append(1, 0, 0.0, False)
    with pytest.raises(AssertionError):                          # nb_entries >= window_length + 2
        small.sample(1)
    small.append(2, 0, 0.0, False)
    assert [e["idx"] for e in small.sample(1, rng=rng)] == [2]   # 3 entries: exactly one experience (entry 1 -> entry 2)


def test_memory_oracle_ring_mapping():
    """valid_transitions(): device ring (slots x lattices) -> the rows keras-rl could sample, for a wrapped and a partial ring."""
    from oracle import memory_oracle as M
    n_envs, n_slots = 3, 8
    term = np.zeros((n_slots, n_envs), np.uint8)
    term[4, 1] = 1                                               # lattice 1 ended its episode with the action taken in slot 4
    # full ring, head (newest observation) in slot 6: entries = slots 7,0,1,2,3,4,5 (oldest first), nb_entries = 7
    rows = M.valid_transitions(term, n_envs, n_slots, 6, 8)
    slots = lambda e: sorted(r // n_envs for r in rows if r % n_envs == e)
    assert slots(0) == [0, 1, 2, 3, 4]                           # not 7 (entry 0), not 5 (newest transition), not 6 (no action yet)
    assert slots(1) == [0, 1, 2, 3, 4]                           # slot 5 would also be excluded as post-terminal -- it is the newest anyway
    term[2, 2] = 1
    rows = M.valid_transitions(term, n_envs, n_slots, 6, 8)
    assert slots(2) == [0, 1, 2, 4]                              # slot 3 holds the terminal observation of the finished episode
    # partial ring: 5 slots written (0..4), head 4 -> entries = slots 0..3; sampleable: slots 1, 2
    rows = M.valid_transitions(np.zeros((n_slots, n_envs), np.uint8), n_envs, n_slots, 4, 5)
    assert sorted(r // n_envs for r in rows if r % n_envs == 0) == [1, 2]


def test_torch_fp32_learner_matches_float64_oracle():
    """oracle/torch_dqn.py (the CPU baseline's learner: torch-CPU fp32, autograd) == the float64 oracle's update to fp32 round-off."""
    from oracle import torch_dqn
    rng = np.random.RandomState(1)
    shape, A = SPECS["c3"]
    spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A)
    flat = O.glorot_init(spec, (5, 6)) + (rng.randn(spec.n_params) * 0.02).astype(np.float32)
    B = 24
    s0, s1 = ((rng.rand(B, *shape) < 0.3).astype(np.uint8) for _ in range(2))
    a, r, term = rng.randint(0, A, B), (rng.rand(B) < 0.5).astype(np.float32), (rng.rand(B) < 0.2).astype(np.uint8)
    keep = rng.rand(B, 512) >= 0.2
    learner = torch_dqn.TorchDQN(spec, flat, lr=1e-3)
    assert np.array_equal(learner.flat(), flat.astype(np.float64))                  # HWIO <-> OIHW round trip
    q = learner.forward(learner.params, s0).detach().numpy()
    assert np.abs(q - O.forward(spec, flat, s0)[0]).max() < 2e-5
    loss, mean_q, grads = learner.update(s0, a, r, term, s1, keep)
    f64 = flat.astype(np.float64)
    y = O.td_targets(O.forward(spec, f64, s1)[0], O.forward(spec, f64, s1)[0], r, term, 0.99)
    q0, cache = O.forward(spec, f64, s0, training=True, keep_masks=[keep])
    loss_ref, mq_ref, dq = O.loss_and_grad(q0, a, y)
    g_ref = O.backward(spec, f64, cache, dq)
    p_ref, _, _ = O.adam_step(f64, g_ref, np.zeros_like(g_ref), np.zeros_like(g_ref), 1, 1e-3)
    assert abs(loss - loss_ref) < 1e-5 and abs(mean_q - mq_ref) < 1e-5
    g = learner.flat(list(grads))
    assert np.abs(g - g_ref).max() < 1e-5 * max(1.0, np.abs(g_ref).max())
    big = np.abs(g_ref) > 1e-5
    assert np.abs(learner.flat() - p_ref)[big].max() < 2e-5


@pytest.mark.parametrize("name", ["c1", "c3"])
def test_dueling_layer_folds_into_one_matrix(name):
    """The identities the HIP path rests on (csrc/qnet.h w3q / wc), checked on the oracle: keras-rl's dueling layer
    Dense(|A| + 1) and its Lambda Q = V + A - mean(A) sit behind the model's last LINEAR layer, so with W3' = W3 C^T, b3' = C b3 (C the
    combination) the forward is Q = y2 W3' + b3'; the backward of both is gY2 = dq W3'^T; and where dq has one non-zero per row (the TD
    loss) gY2[b] = dq[b, a_b] W3'^T[a_b] and gH1[b] = dq[b, a_b] (W3'^T W2^T)[a_b] * mask."""
    shape, A = SPECS[name]
    spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A, dueling=True)
    rng = np.random.RandomState(3)
    flat = O.glorot_init(spec, (5, 6)).astype(np.float64) + rng.randn(spec.n_params) * 0.05
    obs = (rng.rand(9, *shape) < 0.3).astype(np.uint8)
    keep = rng.rand(9, 512) < 0.8
    q, cache = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    P = spec.split(flat)
    (W2, b2), (W3, b3) = P[-2], P[-1]
    fold = lambda M: M[..., 0:1] + M[..., 1:] - M[..., 1:].mean(axis=-1, keepdims=True)      # the combination along the last axis
    W3q, b3q = fold(W3), fold(b3[None, :])[0]
    y2 = cache["layers"][-1]["x"]                                                          # Dense(|A|)'s output = the dueling layer's input
    assert np.abs(y2 @ W3q + b3q - q).max() < 1e-12
    # backward: one non-zero per row of dq
    a_b = rng.randint(0, A, size=9)
    s = rng.randn(9)
    dq = np.zeros((9, A))
    dq[np.arange(9), a_b] = s
    g3 = np.concatenate([dq.sum(axis=1, keepdims=True), dq - dq.sum(axis=1, keepdims=True) / A], axis=1)
    gy2 = g3 @ W3.T
    assert np.abs(gy2 - dq @ W3q.T).max() < 1e-12
    assert np.abs(gy2 - s[:, None] * W3q.T[a_b]).max() < 1e-12
    Wc = W3q.T @ W2.T                                                                    # [|A|, 512]
    D1 = cache["layers"][-3]                                                             # Dense(512): ReLU + dropout
    mask = np.where(keep, 1.0 / 0.8, 0.0) * (D1["y"] > 0.0)
    gh1 = (gy2 @ W2.T) * mask
    assert np.abs(gh1 - s[:, None] * Wc[a_b] * mask).max() < 1e-12
    # Wc by the pack kernel's route: the plain product's row, folded along the row
    assert np.abs(Wc - fold(W2 @ W3).T).max() < 1e-12


def test_dueling_mean_row_and_batch_forms():
    """The one point of the update rule the tree cannot pin (oracle/dqn_oracle.py QNetSpec.dueling_mean): the dueling head's mean per
    sample ("row": Wang et al. eq. 9, keras-rl master, what the HIP kernels fold into W3') or over batch AND actions ("batch": upstream keras-rl 0.4.2's
    axis-less K.mean).  Checked here: (a) both oracle forms equal torch autograd of their own definition; (b) batch-of-one forwards are IDENTICAL -- every
    acting / test forward and every shipped-agent check is form-independent; (c) at the reference's minibatch of 32 the forms differ by a per-batch shift of
    the Q-values (argmax over actions unchanged within a sample) and by a gradient difference that is printed, not hidden."""
    shape, A = SPECS["c3"]
    row = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A, dueling_mean="row")
    bat = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A, dueling_mean="batch")
    rng = np.random.RandomState(11)
    flat = O.glorot_init(row, (5, 6)).astype(np.float64) + rng.randn(row.n_params) * 0.01
    B = 32
    obs = (rng.rand(B, *shape) < 0.3).astype(np.float64)
    q_row, c_row = O.forward(row, flat, obs)
    q_bat, c_bat = O.forward(bat, flat, obs)
    # (b) batch of one
    for b in range(4):
        q1r, _ = O.forward(row, flat, obs[b:b + 1])
        q1b, _ = O.forward(bat, flat, obs[b:b + 1])
        assert np.array_equal(q1r, q1b) and np.allclose(q1r[0], q_row[b], atol=1e-12)
    # (c) the batch form shifts every sample's Q-values by (its own advantage mean - the batch's): same greedy action, different TD targets
    adv = c_row["head_in"][:, 1:]
    shift = adv.mean(axis=1, keepdims=True) - adv.mean()
    assert np.allclose(q_bat, q_row + shift, atol=1e-12)
    assert np.array_equal(q_bat.argmax(axis=1), q_row.argmax(axis=1))
    # (a) + the gradient difference of one 32-sample update with the same targets
    action = rng.randint(0, A, size=B)
    y = rng.randn(B)
    grads = {}
    for name, spec, q, cache in (("row", row, q_row, c_row), ("batch", bat, q_bat, c_bat)):
        _, _, dq = O.loss_and_grad(q, action, y)
        g = O.backward(spec, flat, cache, dq)
        tp = torch.tensor(flat, dtype=torch.float64, requires_grad=True)
        tq = torch_forward(spec, tp, torch.tensor(obs))
        assert np.allclose(tq.detach().numpy(), q, atol=1e-12)
        (tq * torch.tensor(dq)).sum().backward()
        assert np.allclose(g, tp.grad.numpy(), atol=1e-12), name
        grads[name] = g
    diff = np.abs(grads["row"] - grads["batch"]).max()
    rel = np.linalg.norm(grads["row"] - grads["batch"]) / np.linalg.norm(grads["row"])
    print(f"dueling mean, B = 32, random weights: max |Q_batch - Q_row| = {np.abs(shift).max():.3e}; gradient difference max {diff:.3e}, "
          f"relative (2-norm) {rel:.3e}")
    assert diff > 0.0          # the forms ARE different at B = 32: the choice is documented, not immaterial
