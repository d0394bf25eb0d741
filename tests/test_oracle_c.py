"""Pins the plain-C oracle (oracle/env_oracle.c) against the golden traces from the reference and
against the Python restatement.  CPU only."""
import hashlib

import numpy as np
import pytest

from conftest import load_golden, TRACES, STICKY_TRACES, trace_config
from oracle import c_oracle, lattice, philox, referee, env_oracle


def test_c_philox():
    import ctypes
    L = c_oracle.lib()
    for ctr, key in (((0, 0, 0, 0), (0, 0)), ((1, 2, 3, 4), (5, 6)), ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2)):
        c = np.array(ctr, dtype=np.uint32)
        k = np.array(key, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        L.dqo_philox(c.ctypes.data_as(c_oracle._u32p), k.ctypes.data_as(c_oracle._u32p), o.ctypes.data_as(c_oracle._u32p))
        assert tuple(int(x) for x in o) == philox.philox4x32(ctr, key)


@pytest.mark.parametrize("d", [3, 5, 7])
def test_c_lut_matches_golden_hash(d):
    g = load_golden("referee_lut")
    for typ, nm in ((3, "x"), (1, "z")):
        lut = c_oracle.build_lut(d, typ)
        assert hashlib.sha256(lut.tobytes()).digest() == g[f"lut_{nm}_sha256_d{d}"].tobytes()
        if d < 7:
            assert np.array_equal(lut, referee.build_lut(d, typ))


def _obs_from_state(cfg, st, e):
    """Rebuild the (d+1)x(d+1) grids / hidden codes from the exported bit-planes."""
    d = cfg["d"]
    m = lattice.Masks(d)
    hidden = env_oracle.masks_to_codes(d, int(st["xmask"][e]), int(st["zmask"][e]))
    return hidden, m.word_to_grid(int(st["true_word"][e])), m.word_to_grid(int(st["summed"][e]))


def _replay(name, auto_reset):
    g = load_golden("trace_" + name)
    cfg, n_envs, n_steps, seed = trace_config(g)
    env = c_oracle.COracleEnv(n_envs=n_envs, seed=seed, **cfg)
    n_act = env.num_actions

    def check(t):
        st = env.export()
        assert np.array_equal(env.obs, g["obs"][:, t]), (name, t)
        assert np.array_equal(env.done, g["done"][:, t]), (name, t)
        assert np.array_equal(env.lifetime, g["lifetime"][:, t]), (name, t)
        assert np.array_equal(env.legal, g["legal"][:, t]), (name, t)
        assert np.array_equal(st["acted"], g["acted"][:, t])
        assert np.array_equal(st["round"].astype(np.int64), g["rounds"][:, t])
        for e in range(n_envs):
            hidden, true_grid, summed = _obs_from_state(cfg, st, e)
            assert np.array_equal(hidden, g["hidden"][e, t])
            assert np.array_equal(true_grid, g["true_syndrome"][e, t])
            assert np.array_equal(summed, g["summed_nonzero"][e, t])
            comp = (int(st["completed"][e, 0]) | (int(st["completed"][e, 1]) << 64))
            assert [(comp >> a) & 1 for a in range(n_act)] == list(g["completed"][e, t])

    env.reset()
    check(0)
    for t in range(n_steps):
        env.step(g["action"][:, t], auto_reset=auto_reset)
        assert np.array_equal(env.reward, g["reward"][:, t]), (name, t)
        if auto_reset:
            assert np.array_equal(env.was_reset, g["was_reset"][:, t])
        check(t + 1)


@pytest.mark.parametrize("name", TRACES)
def test_c_episode_traces(name):
    _replay(name, True)


@pytest.mark.parametrize("name", STICKY_TRACES)
def test_c_sticky_traces(name):
    _replay(name, False)


@pytest.mark.parametrize("cfg,steps", [
    (dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.02, p_meas=0.02), 150),
    (dict(d=5, error_model="X", use_Y=False, volume_depth=5, p_phys=0.007, p_meas=0.0), 60),       # perfect measurements
    (dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=2e-4, p_meas=2e-4), 25),      # nearly every volume rejected
    (dict(d=5, error_model="DP", use_Y=True, volume_depth=3, p_phys=0.3, p_meas=0.3), 40),         # Y moves, errors everywhere
    (dict(d=5, error_model="IIDXZ", use_Y=False, volume_depth=5, p_phys=0.008, p_meas=0.008), 60),  # independent X and Z flips
    (dict(d=5, error_model="DP", use_Y=True, volume_depth=5, p_phys=0.011, p_meas=0.011), 60),     # Y moves at the headline rates
    (dict(d=7, error_model="X", use_Y=False, volume_depth=7, p_phys=0.006, p_meas=0.006), 40),
    (dict(d=3, error_model="DP", use_Y=False, volume_depth=3, p_phys=0.01, p_meas=0.01), 80),
], ids=["c3-like", "p_meas=0", "rare", "dense-useY", "iidxz", "c3y", "d7x", "d3dp"])
def test_c_vs_python_oracle_random_walk(cfg, steps):
    """Longer free-running cross-check of the two restatements (uniform-over-legal policy), also at the edges of the rate range the GPU
    parity tests use (tests/test_env_gpu.py::test_full_size_vs_c_oracle)."""
    n_envs, seed = 6, (7, 9)
    ce = c_oracle.COracleEnv(n_envs=n_envs, seed=seed, env_id_base=100, **cfg)
    lx, lz = c_oracle.luts(cfg["d"])
    pes = [env_oracle.OracleEnv(referee=referee.LutReferee(cfg["d"], cfg["error_model"], lx, lz), seed=seed, env_id=100 + e, **cfg)
           for e in range(n_envs)]
    ce.reset()
    for p in pes:
        p.reset()
    for t in range(steps):
        a = ce.policy_uniform_legal(t)
        for e, p in enumerate(pes):
            legal = sorted(p.legal_actions)
            w = philox.site_words(seed, 100 + e, t, 0, stream=philox.STREAM_POLICY)
            assert a[e] == legal[philox.bounded(w[0], len(legal))]
        ce.step(a, auto_reset=True)
        for e, p in enumerate(pes):
            if p.done:
                p.reset()
                r = 0.0
            else:
                _, r, _, _ = p.step(int(a[e]))
            assert r == ce.reward[e] and p.done == bool(ce.done[e]) and p.lifetime == ce.lifetime[e]
            assert np.array_equal(p.board_state, ce.obs[e])


def test_partial_reset_and_poke():
    cfg = dict(d=3, error_model="X", use_Y=False, volume_depth=3, p_phys=0.05, p_meas=0.05)
    env = c_oracle.COracleEnv(n_envs=4, **cfg)
    env.reset()
    before = env.export()
    env.reset(which=[0, 1, 0, 1])
    after = env.export()
    assert after["round"][0] == before["round"][0] and after["round"][2] == before["round"][2]
    assert after["round"][1] > before["round"][1] and after["round"][3] > before["round"][3]
    # a logical X (row 0) with the identity action: no anyons, class 1, referee says 0 -> done, reward 0
    env.poke(0, 0b111, 0)
    env.step([9, 9, 9, 9])
    assert env.done[0] == 1 and env.reward[0] == 0.0
