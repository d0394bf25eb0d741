"""GPU parity tests of the HIP Q-network / DQN-update kernels (through the C ABI) against the float64
oracle (oracle/dqn_oracle.py).  Tolerance on Q-values and loss (BASELINE.json north_star "within 1e-5"): scale-aware,
1e-5 * max(1, max |reference|) -- absolute 1e-5 at the unit-scale weights used here (|Q| <~ 2), relative to the largest value at
the trained agents' magnitudes (tests/test_shipped_weights.py); gradients and Adam are checked relative to their scale."""
import numpy as np
import pytest

from oracle import dqn_oracle as O, philox

pytestmark = pytest.mark.gpu

C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
SHAPES = {"c1": ((4, 7, 7), 10), "c2": ((6, 11, 11), 26), "c3": ((7, 11, 11), 51), "c5": ((9, 15, 15), 99),
          # combinations of lattice size and action count beside BASELINE.json's: d = 5 DP with Y moves (use_Y=True: 3 d^2 + 1 = 76 actions on the d = 5 planes),
          # d = 7 X noise (d^2 + 1 = 50 actions on the d = 7 planes), d = 3 DP (2 d^2 + 1 = 19 actions, 5 planes)
          "c3y": ((7, 11, 11), 76), "d7x": ((8, 15, 15), 50), "d3dp": ((5, 7, 7), 19)}


def tol(ref):
    """max(1e-5, 2e-6 max |ref|): absolute 1e-5 (BASELINE.json north_star) up to |Q| = 5, then 2e-6 of the largest value -- five times what the fused
    chains measure on the reference's trained agents (4.2e-7 of max |Q|, tests/test_shipped_weights.py), where no fp32 implementation meets an absolute 1e-5."""
    return max(1e-5, 2e-6 * float(np.abs(np.asarray(ref)).max()))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def _setup(dq, torch, name, batch, seed=(11, 22), dueling=True, max_batch=None, fused=True):
    shape, A = SHAPES[name]
    spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A, dueling=dueling)
    net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, dueling=dueling, max_batch=max_batch or batch)
    assert net.fused_supported                     # the reference architecture runs on the fused chains (csrc/fused*.hip)
    net.set_fused(fused)                           # fused=False: the per-layer implicit-GEMM path (any architecture)
    assert net.n_params == spec.n_params
    params = net.init_params(seed)
    assert np.array_equal(params.cpu().numpy(), O.glorot_init(spec, seed))
    rng = np.random.RandomState(5)
    flat = params.cpu().numpy().copy()
    flat += (rng.randn(flat.size) * 0.02).astype(np.float32)        # non-zero biases, less symmetric weights
    params.copy_(torch.from_numpy(flat))
    obs = (rng.rand(batch, *shape) < 0.3).astype(np.uint8)
    return spec, net, params, flat, obs, rng


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per-layer"])
@pytest.mark.parametrize("name,batch", [("c1", 1), ("c1", 37), ("c2", 32), ("c3", 32), ("c3", 300), ("c5", 64), ("c3y", 70), ("d7x", 45), ("d3dp", 130)])
def test_forward_inference(dq, torch_mod, name, batch, fused):
    torch = torch_mod
    spec, net, params, flat, obs, _ = _setup(dq, torch, name, batch, fused=fused)
    q = net.forward(params, torch.from_numpy(obs).cuda()).cpu().numpy()
    q_ref, _ = O.forward(spec, flat, obs)
    assert np.abs(q - q_ref).max() < tol(q_ref)


def test_forward_non_dueling_and_layer_layout(dq, torch_mod):
    torch = torch_mod
    spec, net, params, flat, obs, _ = _setup(dq, torch, "c3", 16, dueling=False)
    q = net.forward(params, torch.from_numpy(obs).cuda()).cpu().numpy()
    q_ref = O.forward(spec, flat, obs)[0]
    assert np.abs(q - q_ref).max() < tol(q_ref)
    # Keras-shaped views of the flat buffer round-trip (what .h5f loading relies on)
    w = net.get_weights(params)
    assert [x.shape for x in w] == [s for pair in spec.param_shapes() for s in pair]
    for (k, b), wk, wb in zip(spec.split(flat), w[0::2], w[1::2]):
        assert np.array_equal(k, wk) and np.array_equal(b, wb)
    net.set_weights(params, [x * 2 for x in w])
    assert np.array_equal(params.cpu().numpy(), flat * 2)


def test_forward_with_replay_gather(dq, torch_mod):
    """Minibatch rows gathered inside conv1's loader: forward(index) == forward(obs[index]); wrap-around of s1 rows."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 64)
    ring = (rng.rand(200, *SHAPES["c3"][0]) < 0.3).astype(np.uint8)
    idx = rng.randint(0, 200, size=64).astype(np.int32)
    ring_t, idx_t = torch.from_numpy(ring).cuda(), torch.from_numpy(idx).cuda()
    q0 = net.forward(params, ring_t, index=idx_t).cpu().numpy()
    assert np.abs(q0 - O.forward(spec, flat, ring[idx])[0]).max() < tol(q0)
    q1 = net.forward(params, ring_t, index=idx_t, index_off=40, index_mod=200).cpu().numpy()
    assert np.abs(q1 - O.forward(spec, flat, ring[(idx + 40) % 200])[0]).max() < tol(q1)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per-layer"])
@pytest.mark.parametrize("name,batch", [("c1", 8), ("c2", 40), ("c3", 32), ("c3", 257), ("c5", 48), ("c3y", 70), ("d7x", 45), ("d3dp", 130)])
def test_training_forward_backward(dq, torch_mod, name, batch, fused):
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, name, batch, fused=fused)
    seed, t, base = (3, 4), 12345678901, 77
    keep = O.dropout_keep_mask(seed, t, base + np.arange(batch), 512, 0.2)
    q = net.forward(params, torch.from_numpy(obs).cuda(), training=True, seed=seed, t=t, sample_base=base).cpu().numpy()
    q_ref, cache = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    assert np.abs(q - q_ref).max() < tol(q_ref)
    dq_ = (rng.randn(batch, spec.n_actions) / batch).astype(np.float32)
    g = net.backward(params, torch.from_numpy(dq_).cuda()).cpu().numpy()
    g_ref = O.backward(spec, flat, cache, dq_.astype(np.float64))
    scale = np.abs(g_ref).max()
    assert np.abs(g - g_ref).max() < 2e-5 * max(scale, 1.0), (np.abs(g - g_ref).max(), scale)
    # per-layer relative check so that small-gradient layers are not hidden by large ones
    for (gk, gb), (rk, rb) in zip(spec.split(g), spec.split(g_ref)):
        for a, b in ((gk, rk), (gb, rb)):
            assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-7
    # deterministic: same call twice gives identical bits
    g2 = net.backward(params, torch.from_numpy(dq_).cuda()).cpu().numpy()
    assert np.array_equal(g, g2)


@pytest.mark.parametrize("name,batch", [("c3", 4096), ("c3", 77), ("c1", 8)])
def test_dropout_bits_drawn_ahead_equal_the_kernel_own_draw(dq, torch_mod, name, batch):
    """The backward's final reduction draws the keep bits of the training forward it expects next (same seed and sample range, t + 1:
    csrc/qnet.h keep_bits); a forward that asks for exactly that loads them, any other draws in its own kernel -- the same bits, hence the
    same outputs and gradients, bit for bit, and equal to the oracle's mask."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, name, batch)
    _, net2, params2, _, _, _ = _setup(dq, torch, name, batch)
    obs_t = torch.from_numpy(obs).cuda()
    dq_ = torch.from_numpy((rng.randn(batch, spec.n_actions) / batch).astype(np.float32)).cuda()
    seed, t, base = (3, 4), 2 ** 32 - 1, 123                       # t + 1 carries into the counter's high word
    net.forward(params, obs_t, training=True, seed=seed, t=t, sample_base=base)
    net.backward(params, dq_)                                       # draws ahead for (seed, t + 1, base, batch)
    q_ahead = net.forward(params, obs_t, training=True, seed=seed, t=t + 1, sample_base=base).cpu().numpy()
    g_ahead = net.backward(params, dq_).cpu().numpy()
    q_own = net2.forward(params2, obs_t, training=True, seed=seed, t=t + 1, sample_base=base).cpu().numpy()      # no backward before: its own draw
    g_own = net2.backward(params2, dq_).cpu().numpy()
    assert np.array_equal(q_ahead, q_own) and np.array_equal(g_ahead, g_own)
    keep = O.dropout_keep_mask(seed, t + 1, base + np.arange(batch), 512, 0.2)
    q_ref, _ = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    assert np.abs(q_ahead - q_ref).max() < tol(q_ref)
    # a forward the guess does not cover (another sample range) draws for itself
    q_other = net.forward(params, obs_t, training=True, seed=seed, t=t + 2, sample_base=base + 5).cpu().numpy()
    keep = O.dropout_keep_mask(seed, t + 2, base + 5 + np.arange(batch), 512, 0.2)
    assert np.abs(q_other - O.forward(spec, flat, obs, training=True, keep_masks=[keep])[0]).max() < tol(q_ref)


@pytest.mark.parametrize("name,batch", [("c3", 4096), ("c3", 2049), ("c3", 4091), ("c3", 2048 + 8 * 200 + 3), ("c2", 4096), ("c5", 1024), ("c5", 2500),
                                        ("c3", 1500), ("c3", 1021), ("c3y", 4096), ("d7x", 1024), ("d3dp", 4096)])
def test_backward_at_baseline_batch_matches_oracle(dq, torch_mod, name, batch):
    """The fused training forward + backward at the BASELINE.json minibatch sizes against the float64 oracle on the FULL batch.  Only
    batches above 2048 give a workgroup of the persistent convolutional backward more than one group of 8 samples (256 workgroups), i.e.
    exercise its cross-group software pipeline (inputs fetched one group ahead, double-buffered a1, weight-gradient accumulators kept in
    registers across groups) and the batch-slice map of the dense weight-gradient kernel; ragged sizes leave a partly filled last group
    and workgroups with different group counts.  The sizes below 2048 are those where the small-minibatch forms take over: 1500 samples = 94 row
    tiles of the dense backward, two workgroups each (col_split 2: half of gX's column tiles per workgroup); 1021 (and c5's 1024) = 64 row tiles, four
    workgroups each, and groups of 4 instead of 8 samples in the convolutional backward, the last one ragged."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, name, batch)
    seed, t, base = (3, 4), 99, 12345
    keep = O.dropout_keep_mask(seed, t, base + np.arange(batch), 512, 0.2)
    obs_t = torch.from_numpy(obs).cuda()
    q = net.forward(params, obs_t, training=True, seed=seed, t=t, sample_base=base).cpu().numpy()
    q_ref, cache = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    assert np.abs(q - q_ref).max() < tol(q_ref)
    dq_ = (rng.randn(batch, spec.n_actions) / batch).astype(np.float32)
    # A ReLU pre-activation within fp32 round-off of 0 may fall on the other side in another summation order, which changes that
    # sample's gradient by a finite amount; with ~3000 units x thousands of samples some always are.  Such samples (a few per cent)
    # get dq = 0: they still run through every kernel, but their masks cannot matter.
    # (round 6: the threshold halved to 1e-6 -- five times the HIP paths' measured pre-activation error on unit-scale weights -- and the bound set at the measured
    # share: 1.1-2.9 % of a minibatch on these random weights (the largest: d = 7, X noise), where 2e-6 zeroed 2.6-4.3 %; profiles/r06_test_printed_lines.txt)
    fragile = O.fragile_samples(cache, thr=1e-6)
    print(f"{name} B={batch}: {fragile.mean():.3%} of the samples have a ReLU pre-activation within 1e-6 of 0")
    assert fragile.mean() < 0.032
    dq_[fragile] = 0.0
    dq_t = torch.from_numpy(dq_).cuda()
    g = net.backward(params, dq_t).cpu().numpy()
    g_ref = O.backward(spec, flat, cache, dq_.astype(np.float64))
    assert np.abs(g - g_ref).max() < 1e-5 * max(np.abs(g_ref).max(), 1.0), (np.abs(g - g_ref).max(), np.abs(g_ref).max())
    for li, ((gk, gb), (rk, rb)) in enumerate(zip(spec.split(g), spec.split(g_ref))):
        for a, b in ((gk, rk), (gb, rb)):
            assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-7, (li, np.abs(a - b).max(), np.abs(b).max())
    # a gradient that lives in ONE late group only (samples of the last 8-sample group of a late workgroup round): a dropped or stale
    # group cannot hide behind the sum
    one = np.zeros_like(dq_)
    lo = (batch - 1) // 8 * 8
    one[lo:] = dq_[lo:] * 50 + 0.01
    one[fragile] = 0.0
    if not one.any():                                                    # (every sample of the last group fragile: take the group before)
        one[lo - 8:lo] = 0.01
        one[fragile] = 0.0
    g1 = net.backward(params, torch.from_numpy(one).cuda()).cpu().numpy()
    g1_ref = O.backward(spec, flat, cache, one.astype(np.float64))
    assert np.abs(g1 - g1_ref).max() < 1e-5 * max(np.abs(g1_ref).max(), 1.0)
    assert np.abs(g1_ref).max() > 0
    # the per-layer implicit-GEMM path on the same batch (independent kernels) agrees with the oracle too
    net.set_fused(False)
    net.forward(params, obs_t, training=True, seed=seed, t=t, sample_base=base)
    g_pl = net.backward(params, dq_t).cpu().numpy()
    assert np.abs(g_pl - g_ref).max() < 1e-5 * max(np.abs(g_ref).max(), 1.0)


def test_packed_weights_are_equivalent_and_must_follow_the_parameters(dq, torch_mod):
    """forward(packed=net.pack(params)) == forward(params) (which packs on every call), bit for bit; the pack is a pure function
    of the parameters (a stale pack gives the old weights' convolutions, which is why DQNCore repacks after every Adam step)."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 64)
    obs_t = torch.from_numpy(obs).cuda()
    pk = net.pack(params)
    assert pk is not None and pk.numel() == net.packed_bytes
    q_auto = net.forward(params, obs_t)
    q_pk = net.forward(params, obs_t, packed=pk)
    assert torch.equal(q_auto, q_pk)
    p2 = params * 1.5
    q2 = net.forward(p2, obs_t)
    assert not torch.equal(q2, net.forward(p2, obs_t, packed=pk))          # stale pack: wrong on purpose
    assert torch.equal(q2, net.forward(p2, obs_t, packed=net.pack(p2, out=pk)))


def test_forward_multi_equals_separate_forwards(dq, torch_mod):
    """dq_qnet_forward_multi (the update's three forwards in one pair of launches) == three dq_qnet_forward calls, bit for bit;
    the training job's saved activations drive the same backward."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 100, max_batch=128)
    target = params.clone()
    target += 0.01 * torch.randn_like(target)
    ring = torch.from_numpy((rng.rand(500, *SHAPES["c3"][0]) < 0.3).astype(np.uint8)).cuda()
    idx = torch.from_numpy(rng.randint(0, 500, size=100).astype(np.int32)).cuda()
    seed, t, base = (5, 6), 77, 1000
    single = [net.forward(target, ring, index=idx, index_off=60, index_mod=500).cpu().numpy(),
              net.forward(params, ring, index=idx, index_off=60, index_mod=500).cpu().numpy(),
              net.forward(params, ring, index=idx, training=True, seed=seed, t=t, sample_base=base).cpu().numpy()]
    dq_ = torch.from_numpy((rng.randn(100, 51) / 100).astype(np.float32)).cuda()
    g_single = net.backward(params, dq_).cpu().numpy()
    multi = net.forward_multi([dict(params=target, obs=ring, index=idx, index_off=60, index_mod=500),
                               dict(params=params, obs=ring, index=idx, index_off=60, index_mod=500),
                               dict(params=params, obs=ring, index=idx, training=True, seed=seed, t=t, sample_base=base)])
    for a, b in zip(single, multi):
        assert np.array_equal(a, b.cpu().numpy())
    assert np.array_equal(g_single, net.backward(params, dq_).cpu().numpy())
    # two training jobs in one launch are refused (one set of saved activations)
    with pytest.raises(dq.DeepQError):
        net.forward_multi([dict(params=params, obs=ring, index=idx, training=True, seed=seed, t=t)] * 2)


@pytest.mark.parametrize("name,big", [("c3", 1030), ("c5", 333), ("c1", 200), ("c2", 64)])
def test_persistent_conv_forward_equals_the_one_group_kernel(dq, torch_mod, monkeypatch, name, big):
    """The conv forward's persistent form (conv_chain_pkernel: resident workgroups walk the groups of 8 samples, the next group's observations
    prefetched by LDS-DMA a group ahead, a1 planes unpadded) against the one-group-per-workgroup kernel it replaces above 2 workgroups per CU:
    the same arithmetic in the same order, so the Q-values and the gradient computed from the saved activations must be IDENTICAL -- for ragged
    batches, several jobs per launch with different weights, a replay-ring gather with wrap-around, and grids from 1 workgroup (every group
    walked by the same workgroup) to one group each.  (DQ_CONV_PERSIST is read per launch: 0 never, 2 always; DQ_CONV_PERSIST_GRID workgroups.)"""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, name, big, fused=True)
    params2 = (params + 0.05 * torch.randn_like(params)).contiguous()
    T = 37
    ring = torch.from_numpy((rng.rand(T * 64, *spec.input_shape) < 0.3).astype(np.uint8)).cuda()
    index = torch.from_numpy(rng.randint(0, T * 64, size=300).astype(np.int32)).cuda()
    obs_d = torch.from_numpy(obs).cuda()
    dq_ = torch.from_numpy((rng.randn(big, spec.n_actions) / big).astype(np.float32)).cuda()

    def run():
        pk1, pk2 = net.pack(params), net.pack(params2)
        jobs = [dict(params=params, obs=obs_d[:min(250, big)], packed=pk1), dict(params=params2, obs=obs_d[:9], packed=pk2),
                dict(params=params, obs=obs_d, training=True, seed=(5, 6), t=77, packed=pk1),
                dict(params=params2, obs=ring, index=index[:min(300, big)], index_off=T * 64 - 100, index_mod=T * 64, packed=pk2)]
        qs = [q.clone() for q in net.forward_multi(jobs)]
        g = net.backward(params, dq_).clone()
        single = net.forward(params, obs_d[:1]).clone()
        return qs, g, single

    monkeypatch.setenv("DQ_CONV_PERSIST", "0")
    q_ref, g_ref, s_ref = run()
    assert torch.isfinite(g_ref).all() and float(g_ref.abs().max()) > 0
    for grid in ("1", "3", "64", "100000"):
        monkeypatch.setenv("DQ_CONV_PERSIST", "2")
        monkeypatch.setenv("DQ_CONV_PERSIST_GRID", grid)
        qs, g, single = run()
        for a, b in zip(qs, q_ref):
            assert torch.equal(a, b), grid
        assert torch.equal(g, g_ref) and torch.equal(single, s_ref), grid


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per-layer"])
def test_backward_in_two_phases_equals_one_call(dq, torch_mod, fused):
    """dq_qnet_backward_phase 0 (dense) then 1 (convolutions) == dq_qnet_backward, bit for bit; after phase 0 the dense range of the
    gradient is already final (it is all-reduced while phase 1 runs on several GPUs)."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 100, fused=fused)
    dq_ = torch.from_numpy((rng.randn(100, 51) / 100).astype(np.float32)).cuda()
    net.forward(params, torch.from_numpy(obs).cuda(), training=True, seed=(1, 2), t=3)
    g_ref = net.backward(params, dq_).cpu().numpy()
    g = torch.full_like(params, float("nan"))
    net.backward_phase(params, dq_, g, 0)
    nconv = net.n_conv_params
    assert nconv == spec.n_params - sum(int(np.prod(k)) + int(np.prod(b)) for k, b in spec.param_shapes()[3:])
    assert np.array_equal(g[nconv:].cpu().numpy(), g_ref[nconv:]) and torch.isnan(g[:nconv]).all()
    net.backward_phase(params, dq_, g, 1)
    assert np.array_equal(g.cpu().numpy(), g_ref)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per-layer"])
def test_backward_adam_equals_backward_then_adam(dq, torch_mod, fused):
    """dq_qnet_backward_adam (optimizer step applied by the backward's last launch) == dq_qnet_backward + dq_adam_step, bit for bit:
    gradient, parameters and both moment vectors, over three consecutive updates."""
    torch = torch_mod
    from importlib import import_module
    Q = import_module("deepq-decoding_amd.qnet")
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 100, fused=fused)
    obs_t = torch.from_numpy(obs).cuda()
    dq_ = torch.from_numpy((rng.randn(100, 51) / 100).astype(np.float32)).cuda()
    pa, pb = params.clone(), params.clone()
    ma, va, mb, vb = (torch.zeros_like(params) for _ in range(4))
    ga = torch.empty_like(params)
    for t in (1, 2, 3):
        net.forward(pa, obs_t, training=True, seed=(1, 2), t=t)
        gb = net.backward(pb, dq_)
        Q.adam_step(pb, gb, mb, vb, t, 1e-3)
        net.forward(pa, obs_t, training=True, seed=(1, 2), t=t)
        net.backward_adam(pa, dq_, ga, ma, va, t, 1e-3)
        for x, y in ((ga, gb), (pa, pb), (ma, mb), (va, vb)):
            assert torch.equal(x, y)
    assert not torch.equal(pa, params)


def test_td_backward_adam_without_a_dueling_layer(dq, torch_mod):
    """A network without the dueling layer has nothing to fold: its fused TD launch takes the general dense backward and equals the separate calls
    (TD kernel + backward + Adam) bit for bit; the gradient matches the oracle."""
    torch = torch_mod
    from importlib import import_module
    Q = import_module("deepq-decoding_amd.qnet")
    B, A, R = 40, 51, 300
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", B, dueling=False)
    cu = lambda a: torch.from_numpy(a).cuda()
    obs_t = cu(obs)
    q1o, q1t = (cu(rng.randn(B, A).astype(np.float32)) for _ in range(2))
    reward, terminal = cu((rng.rand(R) < 0.4).astype(np.float32)), cu((rng.rand(R) < 0.2).astype(np.uint8))
    action, idx = cu(rng.randint(0, A, size=R).astype(np.int32)), cu(rng.randint(0, R, size=B).astype(np.int32))
    out = {}
    for name in ("separate", "one"):
        p_, m_, v_ = params.clone(), torch.zeros_like(params), torch.zeros_like(params)
        g_ = torch.empty_like(params)
        met = torch.zeros(Q.TD_METRICS_FLOATS, dtype=torch.float32, device="cuda")
        y, dq_ = torch.empty(B, device="cuda"), torch.empty((B, A), device="cuda")
        q0 = net.forward(p_, obs_t, training=True, seed=(1, 2), t=1)
        td = dict(q_online_s1=q1o, q_target_s1=q1t, q_s0=q0, reward=reward, terminal=terminal, action=action, gamma=0.99,
                  grad_scale=1.0 / B, index=idx, y=y, dq=dq_, metrics=met)
        if name == "separate":
            Q.td_update(q1o, q1t, q0, reward, terminal, action, 0.99, grad_scale=1.0 / B, index=idx, y=y, dq=dq_, metrics=met)
            net.set_grad_scale(1.0 / B)
            net.backward(p_, dq_, grads=g_)
            net.set_grad_scale(0.0)
            Q.adam_step(p_, g_, m_, v_, 1, 1e-3)
        else:
            net.td_backward_adam(p_, td, g_, m_, v_, 1, 1e-3)
        out[name] = [x.clone() for x in (y, dq_, g_, p_, m_, v_)]
    for a, b in zip(out["separate"], out["one"]):
        assert torch.equal(a, b)
    keep = O.dropout_keep_mask((1, 2), 1, np.arange(B), 512, 0.2)
    _, cache = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    g_ref = O.backward(spec, flat, cache, out["one"][1].cpu().numpy().astype(np.float64))
    g = out["one"][2].cpu().numpy()
    assert np.abs(g - g_ref).max() < 2e-5 * max(np.abs(g_ref).max(), 1.0)


@pytest.mark.parametrize("fused,shape", [(True, "c3"), (False, "c3"), (True, "c5"), (True, "c3y")], ids=["fused", "per-layer", "fused-99-actions", "fused-76-actions"])
def test_td_backward_adam_equals_the_separate_calls(dq, torch_mod, fused, shape):
    """dq_qnet_td_backward_adam == dq_td_update_stats + dq_qnet_backward + dq_adam_step: y, dq and the episode counters bit for bit; loss / mean_q
    to round-off (their partials are summed in another order); gradient, parameters and moments bit for bit on the per-layer path and between the
    two forms of the fused TD launch (one call / the several-GPU phases), and to round-off between the fused TD launch and the separate calls:
    with the TD step fused in, dq has one non-zero per row and the fused backward forms gY2 = dq W3'^T as a scalar times a table row instead of on
    the matrix pipe (csrc/fused_bwd.hip SHORT) -- the f32 MFMA does not round a lone product as the vector ALU's multiply does.  With more than 64
    actions (c5: 99, c3y: 76) the shortcut reads its table rows from L2 and pack_weights_kernel builds Wc in workgroups of its own; the separate
    calls' backward takes the matrix phases (gY2, gH1 on the f16 pipe)."""
    torch = torch_mod
    from importlib import import_module
    Q = import_module("deepq-decoding_amd.qnet")
    B, A, R = 100, SHAPES[shape][1], 700
    spec, net, params, flat, obs, rng = _setup(dq, torch, shape, B, fused=fused)
    cu = lambda a: torch.from_numpy(a).cuda()
    obs_t = cu(obs)
    q1o, q1t = (cu(rng.randn(B, A).astype(np.float32)) for _ in range(2))
    reward, terminal = cu((rng.rand(R) < 0.4).astype(np.float32)), cu((rng.rand(R) < 0.2).astype(np.uint8))
    action, idx = cu(rng.randint(0, A, size=R).astype(np.int32)), cu(rng.randint(0, R, size=B).astype(np.int32))
    n_envs = 900
    done, was_reset = cu((rng.rand(n_envs) < 0.3).astype(np.uint8)), cu((rng.rand(n_envs) < 0.1).astype(np.uint8))
    life, rew = cu(rng.randint(1, 100, size=n_envs).astype(np.uint32)), cu((rng.rand(n_envs) < 0.5).astype(np.float32))
    out = {}
    for name in ("separate", "one"):
        p_, m_, v_ = params.clone(), torch.zeros_like(params), torch.zeros_like(params)
        g_ = torch.empty_like(params)
        stats = torch.zeros(4, dtype=torch.int64, device="cuda")
        met = torch.full((Q.TD_METRICS_FLOATS,), 7.0, dtype=torch.float32, device="cuda")
        y, dq_ = torch.empty(B, device="cuda"), torch.empty((B, A), device="cuda")
        for t in (1, 2):
            q0 = net.forward(p_, obs_t, training=True, seed=(1, 2), t=t)
            td = dict(q_online_s1=q1o, q_target_s1=q1t, q_s0=q0, reward=reward, terminal=terminal, action=action, gamma=0.99,
                      grad_scale=1.0 / B, index=idx, y=y, dq=dq_, metrics=met, step_stats=(done, was_reset, life, rew, n_envs, stats))
            if name == "separate":
                Q.td_update(q1o, q1t, q0, reward, terminal, action, 0.99, grad_scale=1.0 / B, index=idx, y=y, dq=dq_, metrics=met,
                            step_stats=td["step_stats"])
                net.set_grad_scale(1.0 / B)             # the loss scale dq carries: the fused backward then scales as the TD path does
                net.backward(p_, dq_, grads=g_)
                net.set_grad_scale(0.0)
                Q.adam_step(p_, g_, m_, v_, t, 1e-3)
            else:
                net.td_backward_adam(p_, td, g_, m_, v_, t, 1e-3)
            Q.td_metrics(met, B)
        out[name] = [x.clone() for x in (y, dq_, g_, p_, m_, v_, stats, met[:2])]
    def same(a, b, k):
        if fused and 1 <= k <= 5:                      # dq of the second step, gradient, parameters, moments: see the docstring
            err, scale = float((a - b).abs().max()), float(a.abs().max())
            # (parameters: Adam divides by sqrt(v) + 1e-7 -- where a gradient element is itself ~1e-8, a last-bit difference moves the step)
            return err <= (5e-5 if k == 3 else 1e-6 * max(scale, 1e-30))
        return torch.equal(a, b)
    for k, (a, b) in enumerate(zip(out["separate"][:7], out["one"][:7])):
        assert same(a, b, k), k
    # the several-GPU form: TD + phase 0, then phase 1 and the optimizer step as separate calls
    p_, m_, v_ = params.clone(), torch.zeros_like(params), torch.zeros_like(params)
    g_ = torch.empty_like(params)
    stats = torch.zeros(4, dtype=torch.int64, device="cuda")
    met = torch.zeros(Q.TD_METRICS_FLOATS, dtype=torch.float32, device="cuda")
    y, dq_ = torch.empty(B, device="cuda"), torch.empty((B, A), device="cuda")
    for t in (1, 2):
        q0 = net.forward(p_, obs_t, training=True, seed=(1, 2), t=t)
        td = dict(q_online_s1=q1o, q_target_s1=q1t, q_s0=q0, reward=reward, terminal=terminal, action=action, gamma=0.99,
                  grad_scale=1.0 / B, index=idx, y=y, dq=dq_, metrics=met, step_stats=(done, was_reset, life, rew, n_envs, stats))
        net.td_backward_phase0(p_, td, g_)
        net.backward_phase(p_, dq_, g_, 1)
        Q.adam_step(p_, g_, m_, v_, t, 1e-3)
    for a, b in zip(out["one"][:7], (y, dq_, g_, p_, m_, v_, stats)):      # both run the TD launch: bit for bit
        assert torch.equal(a, b)
    assert torch.allclose(out["separate"][7], out["one"][7], rtol=1e-5, atol=1e-7)
    assert out["one"][6].tolist()[3] == 2 * int((was_reset == 0).sum())


def test_fused_and_per_layer_paths_agree(dq, torch_mod):
    """The two HIP forward/backward implementations order their dot products differently: same results to f32 round-off."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 200)
    obs_t = torch.from_numpy(obs).cuda()
    dq_ = torch.from_numpy((rng.randn(200, 51) / 200).astype(np.float32)).cuda()
    out = {}
    for fused in (True, False):
        net.set_fused(fused)
        q = net.forward(params, obs_t).cpu().numpy()
        qt = net.forward(params, obs_t, training=True, seed=(1, 2), t=3).cpu().numpy()
        out[fused] = (q, qt, net.backward(params, dq_).cpu().numpy())
    assert np.abs(out[True][0] - out[False][0]).max() < 2e-6 and np.abs(out[True][1] - out[False][1]).max() < 2e-6
    # a pre-activation within round-off of 0 may take different sides of the ReLU in the two summation orders: compare robustly
    diff = np.abs(out[True][2] - out[False][2])
    assert np.median(diff) < 1e-8 and (diff > 1e-5 * np.abs(out[False][2]).max()).mean() < 0.02


def test_backward_is_refused_on_the_other_path_than_its_forward(dq, torch_mod):
    """Each path's training forward saves its activations in the form ITS backward reads (the fused one mostly as f16 piece planes,
    the per-layer one as f32): switching paths between the two calls is an error, not a silently wrong gradient."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 64)
    obs_t = torch.from_numpy(obs).cuda()
    dq_ = torch.from_numpy((rng.randn(64, 51) / 64).astype(np.float32)).cuda()
    for first in (True, False):
        net.set_fused(first)
        net.forward(params, obs_t, training=True, seed=(1, 2), t=3)
        net.set_fused(not first)
        with pytest.raises(dq.DeepQError):
            net.backward(params, dq_)
        net.set_fused(first)
        assert np.isfinite(net.backward(params, dq_).cpu().numpy()).all()      # the matching path still works


def test_td_target_loss_adam(dq, torch_mod):
    torch = torch_mod
    rng = np.random.RandomState(9)
    B, A, R = 500, 51, 3000
    q1o, q1t, q0 = (rng.randn(B, A).astype(np.float32) for _ in range(3))
    q1o[7, 3] = q1o[7, 9] = q1o[7].max() + 1.0                      # tie -> first maximum
    reward = (rng.rand(R) < 0.4).astype(np.float32)
    terminal = (rng.rand(R) < 0.2).astype(np.uint8)
    action = rng.randint(0, A, size=R).astype(np.int32)
    idx = rng.randint(0, R, size=B).astype(np.int32)
    cu = lambda a: torch.from_numpy(a).cuda()
    y = dq.td_target(cu(q1o), cu(q1t), cu(reward), cu(terminal), 0.99, index=cu(idx)).cpu().numpy()
    y_ref = O.td_targets(q1o.astype(np.float64), q1t.astype(np.float64), reward[idx], terminal[idx], 0.99)
    assert np.abs(y - y_ref).max() < tol(y_ref)
    dq_, metrics = dq.td_loss_grad(cu(q0), cu(action), cu(y), index=cu(idx))
    loss_ref, mq_ref, dq_ref = O.loss_and_grad(q0.astype(np.float64), action[idx], y.astype(np.float64))
    m = metrics.cpu().numpy()
    assert abs(m[0] - loss_ref) < tol(loss_ref) and abs(m[1] - mq_ref) < tol(mq_ref)
    assert np.abs(dq_.cpu().numpy() - dq_ref).max() < 1e-7
    # Adam, three consecutive updates, odd length (exercises the scalar tail)
    n = 10007
    p, g = rng.randn(n).astype(np.float32), rng.randn(n).astype(np.float32)
    pt, mt, vt = cu(p.copy()), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pr, mr, vr = p.astype(np.float64), np.zeros(n), np.zeros(n)
    for t in range(1, 4):
        dq.adam_step(pt, cu(g), mt, vt, t, 1e-3)
        pr, mr, vr = O.adam_step(pr, g.astype(np.float64), mr, vr, t, 1e-3)
        assert np.abs(pt.cpu().numpy() - pr).max() < 1e-6
    assert np.abs(mt.cpu().numpy() - mr).max() < 1e-6 and np.abs(vt.cpu().numpy() - vr).max() < 1e-6


def test_fused_td_update_and_post_step_equal_the_separate_kernels(dq, torch_mod):
    """dq_td_update == dq_td_target + dq_td_loss_grad (bit for bit); dq_post_step == dq_replay_sample + dq_episode_stats."""
    torch = torch_mod
    from importlib import import_module
    qn, lib = import_module("deepq-decoding_amd.qnet"), import_module("deepq-decoding_amd._lib")
    rng = np.random.RandomState(4)
    B, A, R = 333, 51, 2000
    cu = lambda a: torch.from_numpy(a).cuda()
    q1o, q1t, q0 = (cu(rng.randn(B, A).astype(np.float32)) for _ in range(3))
    reward, terminal = cu((rng.rand(R) < 0.4).astype(np.float32)), cu((rng.rand(R) < 0.2).astype(np.uint8))
    action, idx = cu(rng.randint(0, A, size=R).astype(np.int32)), cu(rng.randint(0, R, size=B).astype(np.int32))
    y = dq.td_target(q1o, q1t, reward, terminal, 0.99, index=idx)
    dq_ref, met_ref = dq.td_loss_grad(q0, action, y, grad_scale=0.01, index=idx)
    y2 = torch.empty_like(y)
    met = torch.zeros(qn.TD_METRICS_FLOATS, dtype=torch.float32, device="cuda")
    dq2 = qn.td_update(q1o, q1t, q0, reward, terminal, action, 0.99, grad_scale=0.01, index=idx, y=y2, metrics=met)
    qn.td_metrics(met, B)
    assert torch.equal(y, y2) and torch.equal(dq_ref, dq2) and torch.equal(met_ref[:2], met[:2])
    # post-step
    n_envs, n_slots, head, filled, batch = 64, 20, 7, 20, 500
    term = cu((rng.rand(n_slots, n_envs) < 0.15).astype(np.uint8))
    done, was_reset = cu((rng.rand(n_envs) < 0.3).astype(np.uint8)), cu((rng.rand(n_envs) < 0.1).astype(np.uint8))
    life, rew = cu(rng.randint(1, 100, size=n_envs).astype(np.uint32)), cu((rng.rand(n_envs) < 0.5).astype(np.float32))
    seed, t, base = (8, 9), 55, 1000
    idx_ref = dq.replay_sample(term, n_envs, n_slots, head, filled, batch, seed, t, sample_base=base)
    stats_ref, stats = torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros(4, dtype=torch.int64, device="cuda")
    L, p = lib.lib(), lib.ptr
    lib.check(L.dq_episode_stats(p(done), p(was_reset), p(life), p(rew), n_envs, p(stats_ref), lib.current_stream()))
    idx2 = torch.empty_like(idx_ref)
    lib.check(L.dq_post_step(p(term), n_envs, n_slots, head, filled, batch, qn._seed_arr(seed), t, base, p(idx2), p(done), p(was_reset), p(life),
                             p(rew), n_envs, p(stats), lib.current_stream()))
    assert torch.equal(idx_ref, idx2) and torch.equal(stats_ref, stats)
    # the bookkeeping riding on the TD launch instead (dq_td_update_stats)
    stats3, met3, y3 = torch.zeros(4, dtype=torch.int64, device="cuda"), torch.zeros_like(met), torch.empty_like(y)
    dq3 = qn.td_update(q1o, q1t, q0, reward, terminal, action, 0.99, grad_scale=0.01, index=idx, y=y3, metrics=met3,
                       step_stats=(done, was_reset, life, rew, n_envs, stats3))
    qn.td_metrics(met3, B)
    assert torch.equal(y, y3) and torch.equal(dq_ref, dq3) and torch.equal(met_ref[:2], met3[:2]) and torch.equal(stats_ref, stats3)


def test_replay_sample_rule(dq, torch_mod):
    """The device sampler against the restatement of upstream keras-rl 0.4.2 SequentialMemory (oracle/memory_oracle.py), lattice by
    lattice: every sampled row is an experience sample() can return (never the newest transition, never entry 0, never one whose
    predecessor entry was terminal), and with enough draws the SUPPORT is exactly keras-rl's -- full ring (wrapped) and partially
    filled ring.  First draws are WITHOUT replacement when the candidates suffice (keras-rl: random.sample), redraws and the
    batch > candidates case with replacement (keras-rl's own fallback); every draw equals the bit-exact restatement
    oracle/memory_oracle.py device_replay_rows."""
    torch = torch_mod
    from oracle import memory_oracle as M
    rng = np.random.RandomState(2)
    seed, t, base = (8, 9), 55, 1000
    for n_envs, n_slots, head, filled, batch in ((64, 50, 17, 50, 60000), (64, 50, 49, 50, 60000), (16, 50, 5, 6, 4000), (7, 9, 3, 4, 500),
                                                  (3, 40, 30, 31, 4000)):
        term = (rng.rand(n_slots, n_envs) < 0.15).astype(np.uint8)
        idx = dq.replay_sample(torch.from_numpy(term).cuda(), n_envs, n_slots, head, filled, batch, seed, t, sample_base=base).cpu().numpy()
        assert np.array_equal(idx, M.device_replay_rows(term, n_envs, n_slots, head, filled, batch, seed, t, sample_base=base))
        allowed = M.valid_transitions(term, n_envs, n_slots, head, filled)
        got = set(int(x) for x in idx)
        assert got <= allowed, sorted(got - allowed)[:5]
        assert got == allowed, (len(got), len(allowed))                # batch >> |allowed|: every allowed experience turns up
        # the oracle's own sample() only ever returns members of that set too (its redraw loop included)
        mem = M.lattice_memory(term, 0, n_slots, head, filled)
        if mem.nb_entries - 2 >= 1:
            ex = mem.sample(min(32, mem.nb_entries - 2)) if mem.valid_idxs() else []
            assert all(e["state0"][0] * n_envs in allowed for e in ex)
            assert all(e["state1"][0] == (e["state0"][0] + 1) % n_slots for e in ex)       # successor = next slot (row + n_envs)
        # roughly uniform over the allowed rows
        if batch >= 20 * len(allowed):
            counts = np.bincount(idx, minlength=n_slots * n_envs)[sorted(allowed)]
            assert counts.min() > 0.15 * batch / len(allowed) and counts.max() < 3.0 * batch / len(allowed)
    # several consecutive updates' minibatches in ONE launch (dq_replay_sample_multi: the extra updates of a vector step): row u = update t + u's own draw
    from importlib import import_module
    Q = import_module("deepq-decoding_amd.qnet")
    term = (rng.rand(50, 64) < 0.15).astype(np.uint8)
    term_t = torch.from_numpy(term).cuda()
    multi = Q.replay_sample_multi(term_t, 64, 50, 17, 50, 777, seed, t, 5, sample_base=base).cpu().numpy()
    for u in range(5):
        assert np.array_equal(multi[u], dq.replay_sample(term_t, 64, 50, 17, 50, 777, seed, t + u, sample_base=base).cpu().numpy()), u
    # WITHOUT replacement (keras-rl random.sample) whenever the candidates suffice: no terminals -> every row of a minibatch distinct,
    # up to the whole candidate set (batch == M: a permutation of it); with terminals the first draws that stand are distinct among
    # themselves (only redraws may repeat a row).  Marginally uniform: over many updates a fixed sample position visits the rows evenly.
    for n_envs, n_slots, head, filled, batch in ((64, 50, 17, 50, 3008), (64, 50, 17, 50, 512), (4096, 257, 100, 257, 4096), (5, 9, 3, 7, 20),
                                                  (1, 40, 7, 40, 37), (3, 5, 1, 4, 3)):
        zeros = np.zeros((n_slots, n_envs), np.uint8)
        idx = dq.replay_sample(torch.from_numpy(zeros).cuda(), n_envs, n_slots, head, filled, batch, seed, t, sample_base=base).cpu().numpy()
        assert len(set(idx.tolist())) == batch, (n_envs, n_slots, batch, len(set(idx.tolist())))
        assert np.array_equal(idx, M.device_replay_rows(zeros, n_envs, n_slots, head, filled, batch, seed, t, sample_base=base))
        if batch == (filled - 3) * n_envs:
            assert set(idx.tolist()) == M.valid_transitions(zeros, n_envs, n_slots, head, filled)
        term = (rng.rand(n_slots, n_envs) < 0.15).astype(np.uint8)
        idx = dq.replay_sample(torch.from_numpy(term).cuda(), n_envs, n_slots, head, filled, batch, seed, t, sample_base=base).cpu().numpy()
        first = M.device_replay_rows(zeros, n_envs, n_slots, head, filled, batch, seed, t, sample_base=base)        # the first draws
        stood = idx == first
        assert len(set(idx[stood].tolist())) == int(stood.sum()) and set(idx.tolist()) <= M.valid_transitions(term, n_envs, n_slots, head, filled)
    n_envs, n_slots, head, filled = 8, 20, 5, 20
    zeros = torch.zeros((n_slots, n_envs), dtype=torch.uint8, device="cuda")
    draws = np.stack([dq.replay_sample(zeros, n_envs, n_slots, head, filled, 32, seed, tt).cpu().numpy() for tt in range(1, 1501)])
    cand_rows = sorted(M.valid_transitions(np.zeros((n_slots, n_envs), np.uint8), n_envs, n_slots, head, filled))
    for pos in (0, 13, 31):                                                # 1500 updates over 136 rows: ~11 visits per row
        counts = np.bincount(draws[:, pos], minlength=n_slots * n_envs)[cand_rows]
        assert counts.sum() == 1500 and counts.max() < 30 and (counts == 0).mean() < 0.01
    chi = ((np.bincount(draws.reshape(-1), minlength=n_slots * n_envs)[cand_rows] - 1500 * 32 / len(cand_rows)) ** 2).sum() / (1500 * 32 / len(cand_rows))
    assert chi < 2.0 * len(cand_rows), chi                                 # pooled over positions: no row favoured
    # the newest sampleable transition is the PREVIOUS step's (slot head - 2); the step just taken (head - 1) never is
    n_envs, n_slots, head, filled = 64, 50, 17, 50
    term = np.zeros((n_slots, n_envs), np.uint8)
    idx = dq.replay_sample(torch.from_numpy(term).cuda(), n_envs, n_slots, head, filled, 20000, seed, t).cpu().numpy()
    slots = set(np.unique(idx // n_envs).tolist())
    assert slots == set(range(n_slots)) - {head, head - 1, (head + 1) % n_slots}           # newest obs, newest transition, entry 0
    # deterministic in (seed, t, sample id): a minibatch is the same whichever launch draws it and however it is sharded
    a = dq.replay_sample(torch.from_numpy(term).cuda(), n_envs, n_slots, head, filled, 100, seed, t, sample_base=40).cpu().numpy()
    b = dq.replay_sample(torch.from_numpy(term).cuda(), n_envs, n_slots, head, filled, 60, seed, t, sample_base=80).cpu().numpy()
    assert np.array_equal(a[40:], b)
    with pytest.raises(dq.DeepQError):                                 # keras-rl: nb_entries >= window_length + 2
        dq.replay_sample(torch.from_numpy(term).cuda(), n_envs, n_slots, 2, 3, 10, seed, t)


def test_one_full_update_matches_oracle(dq, torch_mod):
    """One complete DQN update (double-DQN target from s1, training forward on s0, loss, backward, Adam) on c3,
    batch 32, against the float64 oracle: loss / mean_q within 1e-5, updated weights within 1e-6."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 32, max_batch=64)
    B, A, gamma, lr = 32, 51, 0.99, 1e-4
    target = params.clone()
    target += 0.01 * torch.randn_like(target)
    flat_t = target.cpu().numpy()
    s1 = (rng.rand(B, 7, 11, 11) < 0.3).astype(np.uint8)
    reward = (rng.rand(B) < 0.5).astype(np.float32)
    terminal = (rng.rand(B) < 0.2).astype(np.uint8)
    action = rng.randint(0, A, size=B).astype(np.int32)
    cu = lambda a: torch.from_numpy(a).cuda()
    seed, t = (1, 2), 1
    q1o = net.forward(params, cu(s1))
    q1t = net.forward(target, cu(s1))
    y = dq.td_target(q1o, q1t, cu(reward), cu(terminal), gamma)
    obs_t = cu(obs)
    q0 = net.forward(params, obs_t, training=True, seed=seed, t=t)
    dq_, metrics = dq.td_loss_grad(q0, cu(action), y)
    grads = net.backward(params, dq_)
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    dq.adam_step(params, grads, m, v, 1, lr)
    # oracle
    keep = O.dropout_keep_mask(seed, t, np.arange(B), 512, 0.2)
    y_ref = O.td_targets(O.forward(spec, flat, s1)[0], O.forward(spec, flat_t, s1)[0], reward, terminal, gamma)
    q0_ref, cache = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    loss_ref, mq_ref, dq_ref = O.loss_and_grad(q0_ref, action, y_ref)
    g_ref = O.backward(spec, flat, cache, dq_ref)
    p_ref, _, _ = O.adam_step(flat.astype(np.float64), g_ref, np.zeros_like(g_ref), np.zeros_like(g_ref), 1, lr)
    mt = metrics.cpu().numpy()
    assert abs(mt[0] - loss_ref) < tol(loss_ref) and abs(mt[1] - mq_ref) < tol(mq_ref)
    assert np.abs(y.cpu().numpy() - y_ref).max() < tol(y_ref)
    # first Adam step moves every weight by ~lr * sign(g): compare where the gradient is not ~0
    big = np.abs(g_ref) > 1e-6
    assert np.abs(params.cpu().numpy() - p_ref)[big].max() < 1e-6
    assert np.abs(grads.cpu().numpy() - g_ref).max() < 1e-5 * max(1.0, np.abs(g_ref).max())


def test_qnet_at_baseline_size_properties(dq, torch_mod):
    """c3 at the BASELINE.json batch (4096 samples, the shapes bench.py runs): (1) a strided subset of the rows equals the float64
    oracle; (2) every row equals what the same row gives in a small batch (tiling independence: 16- and 32-row dense workgroups, single-
    and four-job launches); (3) a gathered forward is the permutation of the direct one; (4) the weight gradient is linear in dq."""
    torch = torch_mod
    B = 4096
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", B)
    obs_t = torch.from_numpy(obs).cuda()
    q = net.forward(params, obs_t)
    sub = np.arange(0, B, 97)
    q_ref, _ = O.forward(spec, flat, obs[sub])
    assert np.abs(q[sub].cpu().numpy() - q_ref).max() < tol(q_ref)
    q_small = net.forward(params, obs_t[1000:1048].contiguous(), batch=48)
    assert torch.equal(q_small, q[1000:1048])
    # four jobs in one launch (the fused step's shape: 32-row dense workgroups) == four single launches
    outs = net.forward_multi([dict(params=params, obs=obs_t, batch=B) for _ in range(3)] +
                             [dict(params=params, obs=obs_t, batch=B, training=True, seed=(1, 2), t=3)])
    assert all(torch.equal(o, q) for o in outs[:3])
    q_tr = net.forward(params, obs_t, training=True, seed=(1, 2), t=3)
    assert torch.equal(outs[3], q_tr)
    perm = torch.from_numpy(rng.permutation(B).astype(np.int32)).cuda()
    assert torch.equal(net.forward(params, obs_t, index=perm), q[perm.long()])
    d1 = torch.from_numpy((rng.randn(B, 51) / B).astype(np.float32)).cuda()
    d2 = torch.from_numpy((rng.randn(B, 51) / B).astype(np.float32)).cuda()
    g1 = net.backward(params, d1).clone()
    g2 = net.backward(params, d2).clone()
    g12 = net.backward(params, d1 + d2)
    scale = float(g12.abs().max())
    assert float((g12 - (g1 + g2)).abs().max()) < 2e-5 * scale


def test_guarded_adam_step_skips_and_flags_non_finite_elements(dq, torch_mod):
    """dq_qnet_adam_step (the several-GPU branch's optimizer step behind the gradient all-reduce): an element whose gradient is inf / NaN leaves
    its parameter and moments untouched and raises the handle's range flag, every other element takes dq_adam_step's update bit for bit.  The public
    dq_adam_step is Keras' update: a non-finite gradient element PROPAGATES into its parameter and moments (a divergence stays visible)."""
    torch = torch_mod
    from importlib import import_module
    Q = import_module("deepq-decoding_amd.qnet")
    L = import_module("deepq-decoding_amd._lib")
    spec, net, params, flat, obs, rng = _setup(dq, torch, "c3", 8)
    g = torch.from_numpy(rng.randn(net.n_params).astype(np.float32)).cuda()
    bad = torch.tensor([0, 5, 1001, net.n_params - 1], device="cuda")
    g_bad = g.clone()
    g_bad[bad] = torch.tensor([float("inf"), float("nan"), float("-inf"), float("nan")], device="cuda")
    ref_p, ref_m, ref_v = params.clone(), torch.full_like(params, 0.01), torch.full_like(params, 0.02)
    Q.adam_step(ref_p, g, ref_m, ref_v, 3, 1e-3)
    for guarded in (True, False):
        p_, m_, v_ = params.clone(), torch.full_like(params, 0.01), torch.full_like(params, 0.02)
        if guarded:
            net.check_range()
            net.adam_step(p_, g_bad, m_, v_, 3, 1e-3)
            with pytest.raises(dq.DeepQError) as ei:
                net.check_range()
            assert ei.value.status == L.DQ_ERR_RANGE
            net.check_range()                                    # cleared by the report
        else:
            Q.adam_step(p_, g_bad, m_, v_, 3, 1e-3)
        ok = torch.ones(net.n_params, dtype=torch.bool, device="cuda")
        ok[bad] = False
        assert torch.equal(p_[ok], ref_p[ok]) and torch.equal(m_[ok], ref_m[ok]) and torch.equal(v_[ok], ref_v[ok])
        if guarded:
            assert torch.equal(p_[bad], params[bad]) and bool((m_[bad] == 0.01).all()) and bool((v_[bad] == 0.02).all())
        else:
            assert not bool(torch.isfinite(p_[bad]).any()) and not bool(torch.isfinite(m_[bad]).any())
        assert bool(torch.isfinite(p_).all()) == guarded


@pytest.mark.parametrize("name,batch,patch", [("c3", 300, False), ("c3", 300, True), ("c5", 64, False), ("c1", 37, False)])
def test_forward_range_guard(dq, torch_mod, name, batch, patch):
    """the fused FORWARD carries activations as f16 pieces (finite below 65504); one that leaves that range becomes inf, the products it enters NaN,
    and the next ReLU (v_max_f32) would turn that into 0 -- finite, wrong Q-values.  Every layer's epilogue therefore tracks what it splits and the packing launch
    checks every parameter (csrc/qnet.h range_track); dq_qnet_range_check reports DQ_ERR_RANGE "[forward]".  Healthy weights: silent.  Per-layer f32 path: no guard
    needed, none raised.  patch: patch-word observations, i.e. conv_wave_kernel at d = 5."""
    torch = torch_mod
    spec, net, params, flat, obs, rng = _setup(dq, torch, name, batch)
    obs_t = torch.from_numpy(obs).cuda()
    if patch:
        env = dq.VectorEnv(n_envs=1, d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
        # observations the environment can produce (the embedding's constant cells set): rebuild them from random data bits
        words = torch.from_numpy(rng.randint(0, 1 << 22, size=(batch, env.patch_stride)).astype(np.int32)).cuda()
        obs_t = env.patch_to_obs(words)
        net.set_patch_input(env.volume_depth, env.patch_stride)
        fwd = lambda p: net.forward_multi([dict(params=p, obs=words, patch=True)])[0]
    else:
        fwd = lambda p: net.forward(p, obs_t)
    q = fwd(params)
    net.check_range()                                               # healthy: nothing raised
    assert bool(torch.isfinite(q).all())
    offs = spec.offsets() if hasattr(spec, "offsets") else None
    shapes = spec.param_shapes()
    starts, o = [], 0
    for (ks, bs) in shapes:
        starts.append((o, int(np.prod(ks))))
        o += int(np.prod(ks)) + int(np.prod(bs))
    cases = {}
    for li, label in ((0, "conv1"), (1, "conv2"), (2, "conv3"), (3, "dense512")):
        p = params.clone()
        k0, kn = starts[li]
        p[k0:k0 + kn] *= 0.9 * 65504.0 / float(p[k0:k0 + kn].abs().max())      # the layer's largest weight at 0.9 of the range: its activations (sums of many) outside
        assert float(p.abs().max()) < 65504.0
        cases[label] = p
    p = params.clone(); p[starts[1][0] + 5] = 7.0e4; cases["a parameter of 70000"] = p
    p = params.clone(); p[starts[3][0] + 9] = float("nan"); cases["a NaN parameter"] = p
    p = params.clone(); p[starts[0][0] + starts[0][1] + 3] = float("inf"); cases["an infinite bias"] = p
    obs_np = obs_t.cpu().numpy()
    n_expected = 0
    for label, p in cases.items():
        # what SHOULD happen, from the float64 oracle: the largest activation any layer produces under these parameters (a scaled layer of a small network --
        # d = 3: 32 inputs to Dense(512) -- may stay inside the range: then the guard must stay silent)
        p_np = p.cpu().numpy().astype(np.float64)
        with np.errstate(all="ignore"):
            _, cache = O.forward(spec, p_np, obs_np)
        act = max(float(np.nanmax(np.abs(C["y"]))) if np.isfinite(C["y"]).any() else np.inf for C in cache["layers"][:4])
        bad_param = not np.all(np.abs(p_np) < 65504.0)
        if not bad_param and 5.5e4 < act < 7.5e4:
            continue                                                # (too close to the boundary to call from float64)
        expect = bad_param or not act < 65504.0
        fwd(p)
        if expect:
            n_expected += 1
            with pytest.raises(dq.DeepQError, match="forward") as ei:
                net.check_range()
                pytest.fail(f"forward range guard silent for: {label} (largest activation {act:.3g})")
            assert ei.value.status == -6 and "[forward]" in str(ei.value), label
        print(f"forward range guard, {name} patch={patch}: {label}: largest activation {act:.3g} -> {'raised' if expect else 'silent'}")
        net.check_range()                                           # reported once, then clear / never raised
    assert n_expected >= 5
    fwd(params)
    net.check_range()
    # the per-layer path computes in f32: the same weights overflow nothing there
    if not patch:
        net.set_fused(False)
        q2 = net.forward(cases["conv2"], obs_t)
        net.check_range()
        assert bool(torch.isfinite(q2).all())
