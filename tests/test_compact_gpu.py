"""GPU parity tests of the COMPACT observation path (include/deepq_hip.h dq_env_patch_output, dq_qnet_set_patch_input): the environment
writes d * d patch words per lattice instead of the padded uint8 image, the first convolution reads them (one K = 32 block from the words'
bits, the embedding's constant cells folded into a per-pixel bias / five shared gradient columns).

Checked here: (1) the words the environment writes are exactly the reference's observation (the golden traces recorded from
/root/reference/example_notebooks/Environments.py: padding_syndrome / padding_actions, ENV:273-314), decoded and encoded; (2) Q-values from
patch words meet the float64 oracle's 1e-5 bound like the uint8 path and agree with it; (3) gradients per layer at 1e-4 against the oracle
and the uint8 path; (4) the whole loop on a compact ring equals the loop on a uint8 ring."""
import numpy as np
import pytest

from conftest import load_golden, TRACES, trace_config
from oracle import dqn_oracle as O

pytestmark = pytest.mark.gpu

C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
# name -> (d, syndrome planes, action planes, actions): BASELINE.json's configurations and the combinations tests/test_qnet_gpu.py adds
GEOM = {"c1": (3, 3, 1, 10), "c2": (5, 5, 1, 26), "c3": (5, 5, 2, 51), "c5": (7, 7, 2, 99), "c3y": (5, 4, 3, 76), "d7x": (7, 7, 1, 50), "d3dp": (3, 3, 2, 19)}


def tol(ref):
    return max(1e-5, 2e-6 * float(np.abs(np.asarray(ref)).max()))


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available()
    return torch


def valid_observations(rng, d, depth, layers, batch, p=0.3):
    """Random observations of the reference's form (ENV:273-314): random syndrome bits at the even-even cells over padding_syndrome's
    decoration, random action bits at the odd-odd cells."""
    n = 2 * d + 1
    x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    static = ((((x == 0) | (x == n - 1)) & (y % 2 == 1)) | (((y == 0) | (y == n - 1)) & (x % 2 == 1)) | ((x % 2 == 1) & (y % 2 == 1) & ((x + y) % 4 == 0))).astype(np.uint8)
    obs = np.zeros((batch, depth + layers, n, n), np.uint8)
    obs[:, :depth] = static
    obs[:, :depth, 0::2, 0::2] = rng.rand(batch, depth, d + 1, d + 1) < p
    obs[:, depth:, 1::2, 1::2] = rng.rand(batch, layers, d, d) < p
    return obs


def _setup(dq, torch, name, batch, seed=(11, 22)):
    d, depth, layers, A = GEOM[name]
    shape = (depth + layers, 2 * d + 1, 2 * d + 1)
    spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A, dueling=True)
    net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, dueling=True, max_batch=batch)
    E = __import__("importlib").import_module("deepq-decoding_amd.env")
    stride = E.patch_stride_words(d)
    net.set_patch_input(depth, stride)
    params = net.init_params(seed)
    rng = np.random.RandomState(5)
    flat = params.cpu().numpy().copy()
    flat += (rng.randn(flat.size) * 0.02).astype(np.float32)        # non-zero biases, less symmetric weights
    params.copy_(torch.from_numpy(flat))
    obs = valid_observations(rng, d, depth, layers, batch)
    patch = E.obs_to_patch(torch.from_numpy(obs), d, depth, layers, stride).cuda().contiguous()
    assert torch.equal(E.patch_to_obs(patch, d, depth, layers).cpu(), torch.from_numpy(obs))
    return spec, net, params, flat, obs, patch, rng


@pytest.mark.parametrize("name", [n for n in TRACES])
def test_environment_patch_words_are_the_reference_observation(dq, torch_mod, name):
    """Replays the traces recorded from the reference's Environments.py with the patch output armed on every launch: the words decode to
    the recorded observation (bit for bit), and equal the encoding of the uint8 observation the same launch wrote."""
    torch = torch_mod
    g = load_golden("trace_" + name)
    cfg, n_envs, n_steps, seed = trace_config(g)
    env = dq.VectorEnv(n_envs=n_envs, seed=seed, **cfg)
    assert env.patch_supported
    patch = torch.zeros((n_envs, env.patch_stride), dtype=torch.int32, device="cuda")
    actions = torch.from_numpy(g["action"].astype(np.int32)).cuda()
    from oracle import patch_words as PW
    d, depth, layers = cfg["d"], cfg["volume_depth"], env.n_action_layers
    env.reset(out_patch=patch)
    for t in range(n_steps + 1):
        if t % 8 == 0 or t == n_steps:          # the oracle's cell-by-cell restatement (oracle/patch_words.py) of the RECORDED observation
            want = PW.words_array(g["obs"][:, t].astype(np.uint8), d, depth, layers, env.patch_stride)
            assert np.array_equal(patch.cpu().numpy()[:, :d * d], want[:, :d * d]), (name, "words vs oracle", t)
        assert np.array_equal(env.patch_to_obs(patch).cpu().numpy(), g["obs"][:, t]), (name, "decoded words", t)
        assert torch.equal(env.obs_to_patch(env.obs)[:, :cfg["d"] ** 2], patch[:, :cfg["d"] ** 2]), (name, "encoded image", t)
        if t < n_steps:
            if t % 3 == 2:                      # words only (the loop's form: no uint8 image is written)
                env.arm_patch_output(patch)
                _lib = __import__("importlib").import_module("deepq-decoding_amd._lib")
                _lib.check(env.L.dq_env_step(env._h, actions[:, t].contiguous().data_ptr(), 1, None, env.reward.data_ptr(), env.done.data_ptr(),
                                             env.legal.data_ptr(), env.lifetime.data_ptr(), env.was_reset.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
                env.obs.copy_(env.patch_to_obs(patch))
            else:
                env.step(actions[:, t].contiguous(), auto_reset=True, out_patch=patch)
            assert np.array_equal(env.reward.cpu().numpy(), g["reward"][:, t]), (name, "reward", t)
    # one call arms ONE launch: the next step leaves the words alone
    before = patch.clone()
    env.step(actions[:, 0].contiguous(), auto_reset=True)
    assert torch.equal(before, patch)
    env.close()


@pytest.mark.parametrize("name,batch", [("c1", 37), ("c2", 32), ("c3", 8), ("c3", 300), ("c3", 4096), ("c5", 64), ("c5", 1024), ("c3y", 70), ("d7x", 45), ("d3dp", 130)])
def test_forward_from_patch_words(dq, torch_mod, name, batch):
    torch = torch_mod
    spec, net, params, flat, obs, patch, rng = _setup(dq, torch, name, batch)
    q_u8 = net.forward(params, torch.from_numpy(obs).cuda()).cpu().numpy()
    q_pw = net.forward_multi([dict(params=params, obs=patch, patch=True)])[0].cpu().numpy()
    q_ref, _ = O.forward(spec, flat, obs)
    assert np.abs(q_pw - q_ref).max() < tol(q_ref), ("patch words vs float64", np.abs(q_pw - q_ref).max())
    assert np.abs(q_u8 - q_ref).max() < tol(q_ref)
    assert np.abs(q_pw - q_u8).max() < tol(q_ref), ("patch words vs uint8 image", np.abs(q_pw - q_u8).max())
    # the workgroup-per-group kernels (DQ_CONV_FORM=group; the default at d = 5 is the wave-private form, csrc/conv_wave.hip): the persistent kernel and
    # the one-group kernel give the same bits (as for the uint8 form), and the same Q-values as the default form to round-off
    import os
    old = {k: os.environ.get(k) for k in ("DQ_CONV_PERSIST",)}
    try:
        net.set_kernel_forms(conv_forward="group")
        os.environ["DQ_CONV_PERSIST"] = "2"
        q_p = net.forward_multi([dict(params=params, obs=patch, patch=True)])[0].cpu().numpy()
        os.environ["DQ_CONV_PERSIST"] = "0"
        q_1 = net.forward_multi([dict(params=params, obs=patch, patch=True)])[0].cpu().numpy()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    assert np.array_equal(q_p, q_1)
    assert np.abs(q_p - q_ref).max() < tol(q_ref)
    assert np.abs(q_p - q_pw).max() < 0.2 * tol(q_ref), ("the two forms of the conv forward", np.abs(q_p - q_pw).max())


def test_forward_from_patch_words_with_replay_gather_and_four_jobs(dq, torch_mod):
    """Minibatch rows gathered from a ring of patch words inside conv1's loader (with the successor wrap-around), four jobs in one launch."""
    torch = torch_mod
    spec, net, params, flat, ring, ring_pw, rng = _setup(dq, torch, "c3", 600)
    idx = rng.randint(0, 600, size=256).astype(np.int32)
    idx_t = torch.from_numpy(idx).cuda()
    params2 = params * 1.01
    flat2 = params2.cpu().numpy()
    jobs = [dict(params=params, obs=ring_pw, index=idx_t, patch=True),
            dict(params=params2, obs=ring_pw, index=idx_t, index_off=40, index_mod=600, patch=True),
            dict(params=params, obs=ring_pw[100:], batch=64, patch=True),
            dict(params=params2, obs=ring_pw, index=idx_t, index_off=599, index_mod=600, patch=True)]
    outs = [o.cpu().numpy() for o in net.forward_multi(jobs)]
    refs = [O.forward(spec, flat, ring[idx])[0], O.forward(spec, flat2, ring[(idx + 40) % 600])[0], O.forward(spec, flat, ring[100:164])[0],
            O.forward(spec, flat2, ring[(idx + 599) % 600])[0]]
    for o, r in zip(outs, refs):
        assert np.abs(o - r).max() < tol(r)
    # each job alone gives the same bits as in the shared launch
    for jb, o in zip(jobs, outs):
        assert np.array_equal(net.forward_multi([jb])[0].cpu().numpy(), o)
    # the two forms do not mix in one launch
    with pytest.raises(Exception):
        net.forward_multi([jobs[0], dict(params=params, obs=torch.from_numpy(ring).cuda(), batch=8)])


@pytest.mark.parametrize("name", ["c3", "c2", "c3y"])
def test_wave_private_conv_forward(dq, torch_mod, monkeypatch, name):
    """csrc/conv_wave.hip (the default conv forward for patch words at d = 5: one workgroup of 16 waves per CU, two samples of one job per wave at a
    time, packed weights in LDS, a job's samples dealt over the workgroups of its weight set) against the float64 oracle and against the
    workgroup-per-group kernels: batches of 1, 2, 3 and odd counts (a wave's second sample missing), four jobs with two weight sets of very different
    sizes, a replay gather with wrap-around, more pairs than waves (several trips per wave) and fewer; the training job's saved activations
    through the gradient the backward computes from them; every job alone gives the bits it gives in the shared launch."""
    torch = torch_mod
    spec, net, params, flat, obs, patch, rng = _setup(dq, torch, name, 4099)
    params2 = (params + 0.03 * torch.randn_like(params)).contiguous()
    flat2 = params2.cpu().numpy()
    obs_t = patch
    idx = rng.randint(0, 4099, size=777).astype(np.int32)
    idx_t = torch.from_numpy(idx).cuda()
    dq_ = torch.from_numpy((rng.randn(333, spec.n_actions) / 333).astype(np.float32)).cuda()
    pk1, pk2 = net.pack(params), net.pack(params2)

    def run():
        jobs = [dict(params=params, obs=obs_t[:4099], packed=pk1, patch=True),
                dict(params=params2, obs=obs_t[:3], packed=pk2, patch=True),
                dict(params=params, obs=obs_t[500:], batch=333, training=True, seed=(5, 6), t=77, packed=pk1, patch=True),
                dict(params=params2, obs=obs_t, index=idx_t, index_off=4099 - 50, index_mod=4099, packed=pk2, patch=True)]
        qs = [q.clone() for q in net.forward_multi(jobs)]
        g = net.backward(params, dq_).clone()
        alone = [net.forward_multi([jb])[0].clone() for jb in (jobs[0], jobs[1], jobs[3])]
        tiny = [net.forward_multi([dict(params=params, obs=obs_t[7:7 + n], packed=pk1, patch=True)])[0].clone() for n in (1, 2, 3, 31, 33)]
        return qs, g, alone, tiny

    net.set_kernel_forms(conv_forward="wave")
    qs, g, alone, tiny = run()
    refs = [O.forward(spec, flat, obs[:4099])[0], O.forward(spec, flat2, obs[:3])[0], None, O.forward(spec, flat2, obs[(idx + 4099 - 50) % 4099])[0]]
    for q, r in zip(qs, refs):
        if r is not None:
            assert np.abs(q.cpu().numpy() - r).max() < tol(r)
    for a, q in zip(alone, (qs[0], qs[1], qs[3])):
        assert torch.equal(a, q)
    for n, q in zip((1, 2, 3, 31, 33), tiny):
        assert torch.equal(q, qs[0][7:7 + n]), n
    net.set_kernel_forms(conv_forward="group")
    qs_g, g_g, _, tiny_g = run()
    for a, b, r in zip(qs, qs_g, refs):
        scale = tol(b.cpu().numpy())
        assert (a - b).abs().max().item() < 0.2 * scale
    # the gradient from the wave form's saved a1 / a2 planes against the one from the group form's
    assert torch.isfinite(g).all() and float(g.abs().max()) > 0
    assert (g - g_g).abs().max().item() <= 2e-6 * max(1.0, float(g_g.abs().max()))
    for (gk, gb), (rk, rb) in zip(spec.split(g.cpu().numpy()), spec.split(g_g.cpu().numpy())):
        for a, b in ((gk, rk), (gb, rb)):
            assert np.abs(a - b).max() <= 1e-5 * np.abs(b).max() + 1e-8


@pytest.mark.parametrize("name,batch", [("c1", 8), ("c2", 40), ("c3", 32), ("c3", 257), ("c3", 4096), ("c5", 48), ("c5", 1024), ("c3y", 70), ("d7x", 45), ("d3dp", 130)])
def test_training_forward_backward_from_patch_words(dq, torch_mod, name, batch):
    torch = torch_mod
    spec, net, params, flat, obs, patch, rng = _setup(dq, torch, name, batch)
    seed, t, base = (3, 4), 12345678901, 77
    keep = O.dropout_keep_mask(seed, t, base + np.arange(batch), 512, 0.2)
    q = net.forward_multi([dict(params=params, obs=patch, patch=True, training=True, seed=seed, t=t, sample_base=base)])[0].cpu().numpy()
    q_ref, cache = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    assert np.abs(q - q_ref).max() < tol(q_ref)
    dq_ = (rng.randn(batch, spec.n_actions) / batch).astype(np.float32)
    g = net.backward(params, torch.from_numpy(dq_).cuda()).cpu().numpy()
    g_ref = O.backward(spec, flat, cache, dq_.astype(np.float64))
    scale = np.abs(g_ref).max()
    assert np.abs(g - g_ref).max() < 2e-5 * max(scale, 1.0), (np.abs(g - g_ref).max(), scale)
    for (gk, gb), (rk, rb) in zip(spec.split(g), spec.split(g_ref)):
        for a, b in ((gk, rk), (gb, rb)):
            assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-7
    # the first kernel's gradient row by row: rows no observation can excite are exactly 0 in both
    k1 = spec.split(g)[0][0].reshape(-1, 64)
    r1 = spec.split(g_ref)[0][0].reshape(-1, 64)
    assert np.array_equal(np.abs(r1).max(axis=1) == 0, np.abs(k1).max(axis=1) == 0)
    # deterministic
    assert np.array_equal(g, net.backward(params, torch.from_numpy(dq_).cuda()).cpu().numpy())
    # ... and the uint8 form of the same observations: same function of the same weights
    q8 = net.forward(params, torch.from_numpy(obs).cuda(), training=True, seed=seed, t=t, sample_base=base).cpu().numpy()
    g8 = net.backward(params, torch.from_numpy(dq_).cuda()).cpu().numpy()
    assert np.abs(q - q8).max() < tol(q_ref)
    for (gk, gb), (rk, rb) in zip(spec.split(g), spec.split(g8)):
        for a, b in ((gk, rk), (gb, rb)):
            assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-7


@pytest.mark.parametrize("name,batch", [("c3", 264), ("c3", 2048), ("c2", 1032), ("c3y", 1024)])
def test_sixteen_wave_conv_backward(dq, torch_mod, monkeypatch, name, batch):
    """csrc/conv_bwd16.hip (patch words at d = 5, minibatches that are multiples of 8; the default from 1024 samples) against the float64 oracle at the
    bounds of the test above, against csrc/fused_bwd.hip's 8-wave kernel on the same saved activations (DQ_CONV_BWD_FORM=8) to round-off, and run to run
    (bit-identical: fixed-order sums)."""
    torch = torch_mod
    spec, net, params, flat, obs, patch, rng = _setup(dq, torch, name, batch)
    seed, t, base = (5, 6), 424242, 11
    keep = O.dropout_keep_mask(seed, t, base + np.arange(batch), 512, 0.2)
    dq_ = (rng.randn(batch, spec.n_actions) / batch).astype(np.float32)
    g = {}
    # "16": the default -- the training forward (conv_wave_kernel) does not save a1, the backward recomputes it from the patch words (round 6); "16saved": the
    # round-5 form, a1 through HBM; "8": fused_bwd.hip's kernel, always on saved planes
    job = dict(params=params, obs=patch, patch=True, training=True, seed=seed, t=t, sample_base=base)
    for form in ("16saved", "16", "8"):
        if form == "16":
            # poison the saved a1 planes first: a training forward in the saving form on OTHER observations (rows reversed) -- the recomputing forward below must
            # leave them as they are, and a backward that still read them would differentiate the wrong activations
            net.set_kernel_forms(conv_backward="16", conv_backward_a1="saved")
            net.forward_multi([dict(job, obs=torch.flip(patch, dims=[0]).contiguous())])
        net.set_kernel_forms(conv_backward=form[:2].rstrip("s"), conv_backward_a1="saved" if form == "16saved" else "recompute")
        net.forward_multi([job])
        g[form] = net.backward(params, torch.from_numpy(dq_).cuda()).cpu().numpy()
        assert np.array_equal(g[form], net.backward(params, torch.from_numpy(dq_).cuda()).cpu().numpy())
    # recomputed a1 == saved a1, bit for bit: identical gradients
    assert np.array_equal(g["16"], g["16saved"])
    # a change of form between a training forward and its backward that leaves the backward without a1 is refused
    net.set_kernel_forms(conv_backward="16", conv_backward_a1="recompute")
    net.forward_multi([dict(params=params, obs=patch, patch=True, training=True, seed=seed, t=t, sample_base=base)])
    net.set_kernel_forms(conv_backward="8")
    with pytest.raises(dq.DeepQError):
        net.backward(params, torch.from_numpy(dq_).cuda())
    net.set_kernel_forms(conv_backward="default")
    _, cache = O.forward(spec, flat, obs, training=True, keep_masks=[keep])
    g_ref = O.backward(spec, flat, cache, dq_.astype(np.float64))
    assert np.abs(g["16"] - g_ref).max() < 2e-5 * max(np.abs(g_ref).max(), 1.0)
    for (gk, gb), (rk, rb), (ok, ob) in zip(spec.split(g["16"]), spec.split(g_ref), spec.split(g["8"])):
        for a, b, c in ((gk, rk, ok), (gb, rb, ob)):
            assert np.abs(a - b).max() <= 1e-4 * np.abs(b).max() + 1e-7
            assert np.abs(a - c).max() <= 1e-5 * np.abs(c).max() + 1e-8
    # rows of the first kernel no observation can excite are exactly 0 in both
    k16, k8 = spec.split(g["16"])[0][0].reshape(-1, 64), spec.split(g["8"])[0][0].reshape(-1, 64)
    assert np.array_equal(np.abs(k16).max(axis=1) == 0, np.abs(k8).max(axis=1) == 0)


@pytest.mark.parametrize("cfg,n,batch", [(dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011), 512, 512),
                                         (dict(d=7, error_model="DP", use_Y=False, volume_depth=7, p_phys=0.005, p_meas=0.005), 128, 128),
                                         (dict(d=3, error_model="X", use_Y=False, volume_depth=3, p_phys=0.005, p_meas=0.005), 1, 32)])
def test_loop_on_a_compact_ring_equals_the_loop_on_a_uint8_ring(dq, torch_mod, cfg, n, batch):
    """DQNCore with the ring as patch words against DQNCore with the uint8 ring (compact=False): same lattices, same minibatches; the
    transitions (actions, rewards, terminals, observations) are identical, the parameters agree to f32 round-off."""
    torch = torch_mod
    seed = (7, 9)
    cores = []
    for compact in (True, False):
        env = dq.VectorEnv(n_envs=n, seed=seed, **cfg)
        net = dq.QNetwork(env.obs_shape, C_LAYERS, FF_LAYERS, env.num_actions, max_batch=max(n, batch))
        core = dq.DQNCore(env, net, batch_size=batch, memory_limit=n * 9, gamma=0.99, lr=1e-4, seed=seed, compact=compact)
        assert core.compact == compact
        core.reset_env()
        for _ in range(5):
            core.act_and_step(1.0, use_q=False)
        cores.append(core)
    a, b = cores
    assert a.patch_ring is not None and b.patch_ring is None
    lr, steps = 1e-4, 12
    for t in range(steps):
        for c in cores:
            c.step_and_update(1.0)              # (eps = 1: the actions do not hang on a round-off-level tie between two Q-values)
        assert torch.equal(a.index, b.index), t
        qa, qb = a.q_act.cpu().numpy(), b.q_act.cpu().numpy()
        assert np.abs(qa - qb).max() < tol(qb), t
    assert torch.equal(a.action_ring, b.action_ring) and torch.equal(a.reward_ring, b.reward_ring) and torch.equal(a.terminal_ring, b.terminal_ring)
    assert torch.equal(a.obs_ring[:], b.obs_ring[:])
    la, lb = a.read_metrics(), b.read_metrics()
    assert abs(la[0] - lb[0]) < 1e-5 * max(1.0, abs(lb[0])) and abs(la[1] - lb[1]) < 1e-5 * max(1.0, abs(lb[1]))
    # Adam divides by sqrt(v): an element whose gradient is at round-off level moves by up to lr per step in either run; the bulk agrees closely
    diff = (a.params - b.params).abs()
    assert float(diff.max()) <= 2 * lr * steps and float(diff.mean()) < 2e-6, (float(diff.max()), float(diff.mean()))
    assert a.read_stats() == b.read_stats()


def test_configurations_beyond_one_word_per_pixel_keep_the_uint8_ring(dq, torch_mod):
    """4 volume_depth + action layers > 32 data bits per pixel (volume_depth 8 at d = 5) or the wide environment (d = 9): DQNCore stays on the
    uint8 ring, dq_env_patch_output refuses, and the loop runs as before."""
    torch = torch_mod
    env = dq.VectorEnv(n_envs=32, d=5, error_model="DP", use_Y=False, volume_depth=8, p_phys=0.01, p_meas=0.01)
    assert not env.patch_supported
    with pytest.raises(dq.DeepQError):
        env.arm_patch_output(torch.zeros((32, env.patch_stride), dtype=torch.int32, device="cuda"))
    net = dq.QNetwork(env.obs_shape, C_LAYERS, FF_LAYERS, env.num_actions, max_batch=32)
    with pytest.raises(dq.DeepQError):
        net.set_patch_input(8, 32)
    core = dq.DQNCore(env, net, batch_size=32, memory_limit=32 * 8, gamma=0.99, lr=1e-4)
    assert not core.compact and core.patch_ring is None and core.obs_ring.dtype == torch.uint8
    core.reset_env()
    for _ in range(4):
        core.act_and_step(1.0, use_q=False)
    for _ in range(3):
        core.step_and_update(0.3)
    loss, mean_q = core.read_metrics()
    assert np.isfinite(loss) and np.isfinite(mean_q)


def test_last_convolution_output_as_piece_planes_changes_no_bit(dq, torch_mod, monkeypatch):
    """DQ_X_PLANES=1 (round 6, off by default: measured no faster): conv_wave_kernel splits the last convolution's output on write and the dense chain stages the
    piece planes by LDS-DMA instead of splitting f32 rows itself -- the same split of the same f32 values, so Q-values and gradients are bit-identical to the
    default path, for inference jobs (planes in the job's scratch buffer), the training job (planes straight into the weight gradients' operand) and ragged batches."""
    torch = torch_mod
    out = {}
    for planes in ("0", "1"):
        monkeypatch.setenv("DQ_X_PLANES", planes)
        spec, net, params, flat, obs, patch, rng = _setup(dq, torch, "c3", 1091)
        seed, t, base = (5, 6), 77, 3
        dq_ = (np.random.RandomState(3).randn(1091, spec.n_actions) / 1091).astype(np.float32)
        q_inf = net.forward_multi([dict(params=params, obs=patch, patch=True), dict(params=params, obs=patch[:37].contiguous(), patch=True)])
        q_inf = [q.clone() for q in q_inf]
        q_tr = net.forward_multi([dict(params=params, obs=patch, patch=True, training=True, seed=seed, t=t, sample_base=base)])[0].clone()
        g = net.backward(params, torch.from_numpy(dq_).cuda()).clone()
        out[planes] = q_inf + [q_tr, g]
    for a, b in zip(out["0"], out["1"]):
        assert torch.equal(a, b)


def test_lean_dense_forward_changes_no_bit(dq, torch_mod, monkeypatch):
    """DQ_DENSE_LEAN=2 (round 6, off by default: measured slower): the dense forward in at most 128 VGPRs, two 32-row workgroups per CU -- one ring of weight tiles
    refilled in place, Dense(|A|)'s weights one K block at a time -- feeds every accumulator the same products in the same order: Q-values, the saved planes the
    backward reads, and therefore the gradients are bit-identical to the default form's, for full, ragged and tiny batches."""
    torch = torch_mod
    out = {}
    for lean in ("0", "2"):
        monkeypatch.setenv("DQ_DENSE_LEAN", lean)
        spec, net, params, flat, obs, patch, rng = _setup(dq, torch, "c3", 1091)
        seed, t, base = (5, 6), 77, 3
        dq_ = (np.random.RandomState(3).randn(1091, spec.n_actions) / 1091).astype(np.float32)
        q_inf = [q.clone() for q in net.forward_multi([dict(params=params, obs=patch, patch=True), dict(params=params, obs=patch[:37].contiguous(), patch=True)])]
        q_one = net.forward_multi([dict(params=params, obs=patch[5:6].contiguous(), patch=True)])[0].clone()
        q_tr = net.forward_multi([dict(params=params, obs=patch, patch=True, training=True, seed=seed, t=t, sample_base=base)])[0].clone()
        g = net.backward(params, torch.from_numpy(dq_).cuda()).clone()
        out[lean] = q_inf + [q_one, q_tr, g]
    for a, b in zip(out["0"], out["2"]):
        assert torch.equal(a, b)
