"""DQN half of __graft_entry__.smoke(): a few vector steps of the full device loop (act, environment step, replay ring, one
double-DQN update per step through the fused Q-network kernels) on a small batch, and one training forward/backward checked
against the float64 oracle (oracle/ is test infrastructure: it is only the checker here)."""
import importlib

import numpy as np
import torch

_dq = importlib.import_module("deepq-decoding_amd")
DQNCore, VectorEnv, QNetwork = _dq.DQNCore, _dq.VectorEnv, _dq.QNetwork

C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]


def run():
    from oracle import dqn_oracle as O
    cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
    env = VectorEnv(n_envs=64, **cfg)
    net = QNetwork(env.obs_shape, C_LAYERS, FF_LAYERS, env.num_actions, dueling=True, max_batch=64)
    assert net.fused_supported, "the reference architecture must run on the fused chains"
    core = DQNCore(env, net, batch_size=64, memory_limit=64 * 16, gamma=0.99, lr=1e-4)
    core.reset_env()
    for _ in range(3):
        core.act_and_step(0.1)
    for _ in range(2):
        core.act_and_step(0.1)
        core.update()
    for _ in range(2):
        core.step_and_update(0.1)          # the same step with the four forwards in one pair of launches
    loss, mean_q = core.read_metrics()
    assert np.isfinite(loss) and np.isfinite(mean_q)
    # parity of one training forward + backward on the current observations
    spec = O.QNetSpec(env.obs_shape, C_LAYERS, FF_LAYERS, env.num_actions)
    flat = core.params.cpu().numpy()
    obs = core.obs_ring[core.cur]
    seed, t = (3, 4), 9
    q = net.forward(core.params, obs, training=True, seed=seed, t=t).cpu().numpy()
    keep = O.dropout_keep_mask(seed, t, np.arange(64), 512, 0.2)
    q_ref, cache = O.forward(spec, flat, obs.cpu().numpy(), training=True, keep_masks=[keep])
    assert np.abs(q - q_ref).max() < 1e-5, np.abs(q - q_ref).max()
    dq_ = (np.random.RandomState(0).randn(64, env.num_actions) / 64).astype(np.float32)
    g = net.backward(core.params, torch.from_numpy(dq_).cuda()).cpu().numpy()
    g_ref = O.backward(spec, flat, cache, dq_.astype(np.float64))
    assert np.abs(g - g_ref).max() < 2e-5 * max(1.0, np.abs(g_ref).max()), np.abs(g - g_ref).max()
    # the loop's own form of the observation: the ring holds patch words (core.compact) -- the words equal the oracle's cell-by-cell restatement
    # of the decoded image, and the training forward / backward FROM THE WORDS meets the same bounds
    assert core.compact, "the headline configuration runs on the compact ring"
    from oracle import patch_words as PW
    words = core.patch_ring[core.cur]
    want = PW.words_array(obs.cpu().numpy(), env.d, env.volume_depth, env.n_action_layers, env.patch_stride)
    assert np.array_equal(words.cpu().numpy()[:, :env.d ** 2], want[:, :env.d ** 2])
    q2 = net.forward_multi([dict(params=core.params, obs=words.contiguous(), patch=True, training=True, seed=seed, t=t)])[0].cpu().numpy()
    assert np.abs(q2 - q_ref).max() < 1e-5, np.abs(q2 - q_ref).max()
    g2 = net.backward(core.params, torch.from_numpy(dq_).cuda()).cpu().numpy()
    assert np.abs(g2 - g_ref).max() < 2e-5 * max(1.0, np.abs(g_ref).max()), np.abs(g2 - g_ref).max()
    print(f"agent smoke ok: loss {loss:.4g} mean_q {mean_q:.4g}")
