"""CPU-only tests of the host-side mirror of the reference interface: the Function_Library-named helpers against
the golden vectors, policies / schedules / logging / memory pickling, and the drop-in module tree."""
import importlib
import json
import os
import pickle
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import dqn_oracle as O


@pytest.fixture(scope="module")
def fl():
    return importlib.import_module("deepq-decoding_amd.function_library")


@pytest.fixture(scope="module")
def agent_mod():
    return importlib.import_module("deepq-decoding_amd.agent")


@pytest.mark.parametrize("d", [3, 5, 7])
def test_function_library_vs_golden(fl, d):
    g, t = load_golden("kats"), load_golden("tables")
    qubits = fl.generateSurfaceCodeLattice(d)
    assert np.array_equal(qubits, t[f"qubits_d{d}"])
    errs = g[f"kat_err_d{d}"].astype(int)
    for i in range(0, len(errs), 7):
        assert np.array_equal(fl.generate_surface_code_syndrome_NoFT_efficient(errs[i], qubits), g[f"kat_syn_d{d}"][i])
        assert np.array_equal(fl.generate_one_hot_labels_surface_code(errs[i], "DP"), g[f"kat_label_dp_d{d}"][i])
        assert np.array_equal(fl.generate_one_hot_labels_surface_code(errs[i] * (errs[i] == 1), "X"), g[f"kat_label_x_d{d}"][i])
    a, b = errs[-200:-100], errs[-100:]
    for i in range(0, 100, 9):
        out = fl.obtain_new_error_configuration(a[i], b[i])
        assert out.dtype == np.float64 and np.array_equal(out, g[f"kat_prod_d{d}"][i])
    assert [[fl.multiplyPaulis(x, y) for y in range(4)] for x in range(4)] == g[f"kat_mul_d{d}"].tolist()
    for model, use_Y in (("X", False), ("DP", True), ("DP", False)):
        want = g[f"kat_move_d{d}_{model}_{int(use_Y)}"]
        for act in range(want.shape[0]):
            assert np.array_equal(fl.index_to_move(d, act, model, use_Y), want[act])
    # measurement-error draw order: serve the recorded words through np.random.rand
    saved = np.random.rand
    try:
        for i in range(0, 64, 5):
            words = g[f"kat_faulty_words_d{d}"][i]
            np.random.rand = lambda n, w=words: w[:n].astype(np.float64) / 4294967296.0
            out = fl.generate_faulty_syndrome(g[f"kat_faulty_true_d{d}"][i].astype(int), float(g[f"kat_faulty_p_d{d}"][i]))
            assert np.array_equal(out, g[f"kat_faulty_out_d{d}"][i])
    finally:
        np.random.rand = saved


def test_error_generators(fl):
    np.random.seed(0)
    e = fl.generate_error(5, 0.3, "X")
    assert e.shape == (5, 5) and set(np.unique(e)) <= {0, 1}
    e = fl.generate_error(7, 0.5, "DP")
    assert set(np.unique(e)) <= {0, 1, 2, 3} and (e > 0).sum() > 5
    e = fl.generate_error(5, 0.5, "IIDXZ")
    assert set(np.unique(e)) <= {0, 1, 2, 3}
    assert fl.generate_error(5, 0.0, "DP").sum() == 0
    # the host helper generate_IIDXZ_error consumes numpy's stream like the reference's loop (two uniforms per qubit, X first):
    # served the golden words, it returns the reference's own outputs (tests/golden/kats.npz kat_iidxz_*)
    g = load_golden("kats")
    saved = np.random.rand
    try:
        for d in (3, 5, 7):
            for w, want in zip(g[f"kat_iidxz_words_d{d}"], g[f"kat_iidxz_err_d{d}"]):
                stream = iter(w.astype(np.float64) / 4294967296.0)
                np.random.rand = lambda *shape: (np.array([next(stream) for _ in range(int(np.prod(shape)))]).reshape(shape)
                                                 if shape else next(stream))
                assert np.array_equal(fl.generate_error(d, 0.3, "IIDXZ"), want)
    finally:
        np.random.rand = saved
    with pytest.raises(Exception):
        fl.generateSurfaceCodeLattice(4)


def test_policies_and_schedule(agent_mod):
    A = agent_mod

    class FakeAgent:
        step = 0
    pol = A.LinearAnnealedPolicy(A.EpsGreedyQPolicy(masked_greedy=False), attr="eps", value_max=1.0, value_min=0.02, value_test=0.0,
                                 nb_steps=100000)
    fa = FakeAgent()
    pol._set_agent(fa)
    for step in (0, 1, 50000, 99999, 100000, 10 ** 7):
        fa.step = step
        eps, masked = pol.current(True)
        assert eps == O.annealed_eps(step, 1.0, 0.02, 100000) and masked is False
    assert pol.current(False) == (0.0, False)
    assert pol.metrics_names == ["mean_eps"]
    assert A.GreedyQPolicy(masked_greedy=True).current(False) == (0.0, True)
    with pytest.raises(ValueError):
        A.LinearAnnealedPolicy(A.GreedyQPolicy(), "eps", 1, 0, 0, 10)
    with pytest.raises(NotImplementedError):
        A.BoltzmannQPolicy()
    opt = A.Adam(lr=1e-4)
    assert (opt.lr, opt.beta_1, opt.beta_2, opt.epsilon) == (1e-4, 0.9, 0.999, 1e-7)


def test_agent_argument_checks(agent_mod):
    A = agent_mod
    model = A.build_convolutional_nn([[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], (7, 11, 11), 51)
    assert model.output_shape == (None, 51)
    assert model.layer_names() == ["conv2d_1", "conv2d_2", "conv2d_3", "dense_1", "dense_2", "dense_3"]
    mem = A.SequentialMemory(limit=50000, window_length=1)
    with pytest.raises(ValueError):
        A.DQNAgent(model=model, nb_actions=50, memory=mem)
    with pytest.raises(NotImplementedError):
        A.SequentialMemory(limit=10, window_length=4)
    dqn = A.DQNAgent(model=model, nb_actions=51, memory=mem, nb_steps_warmup=1000, target_model_update=5000,
                     policy=A.EpsGreedyQPolicy(), test_policy=A.GreedyQPolicy(masked_greedy=True), gamma=0.99,
                     enable_dueling_network=True)
    assert dqn.metrics_names == ["loss", "mean_q"] and dqn.get_config()["enable_double_dqn"] is True
    with pytest.raises(RuntimeError):
        dqn.forward(np.zeros((7, 11, 11)))
    mem2 = pickle.loads(pickle.dumps(mem))
    assert mem2.limit == 50000 and mem2.nb_entries == 0


def test_rolling_window_and_filelogger(agent_mod, tmp_path):
    A = agent_mod
    from collections import deque
    dq_ = deque()
    for life in (10, 20, 30, 40):
        A.DQNAgent._append_lifetimes(dq_, 1, life, 3)
    assert sum(c for c, _ in dq_) == 3 and sum(s for _, s in dq_) == 90
    dq_ = deque()
    A.DQNAgent._append_lifetimes(dq_, 100, 1000, 50)       # one big chunk is kept whole
    A.DQNAgent._append_lifetimes(dq_, 60, 1200, 50)
    assert list(dq_) == [(60, 1200)]
    path = tmp_path / "training_history.json"
    log = A.FileLogger(str(path), interval=2)
    for ep in range(5):
        log.on_episode_end(ep, {"episode_reward": float(ep), "loss": float("nan") if ep < 2 else 0.1, "nb_steps": 10 * ep})
    log.on_train_end()
    data = json.loads(path.read_text())
    assert data["episode"] == [0, 1, 2, 3, 4] and data["nb_steps"][-1] == 40 and data["loss"][0] != data["loss"][0]


def test_weights_file_roundtrip(agent_mod, tmp_path):
    wio = importlib.import_module("deepq-decoding_amd.weights_io")
    rng = np.random.RandomState(0)
    names = ["conv2d_1", "dense_1"]
    w = [rng.randn(3, 3, 7, 64).astype(np.float32), rng.randn(64).astype(np.float32),
         rng.randn(288, 512).astype(np.float32), rng.randn(512).astype(np.float32)]
    p = str(tmp_path / "final_dqn_weights.h5f")
    wio.save_weights_file(p, w, names)
    back = wio.load_weights_file(p)
    assert len(back) == 4 and all(np.array_equal(a, b) for a, b in zip(w, back))


def test_hdf5_writer_produces_the_shipped_files_structure(tmp_path):
    """save_weights writes a Keras 2.x HDF5 file: the shipped agent's tensors (golden fixture) survive the trip bit for bit, the tree
    and the attributes Keras' load_weights reads (layer_names, weight_names, keras_version, backend) are as in the shipped files,
    and the container obeys the format's bookkeeping (superblock, end-of-file address, 8-byte alignment, sorted symbol tables)."""
    import struct
    h = importlib.import_module("deepq-decoding_amd.hdf5_reader")
    wio = importlib.import_module("deepq-decoding_amd.weights_io")
    fx = load_golden("keras_weights_d5_dp_0.007")
    w = [fx[f"w{i}"] for i in range(12)]
    names = ["conv2d_1", "conv2d_2", "conv2d_3", "dense_1", "dense_2", "dense_3"]
    p = str(tmp_path / "final_dqn_weights.h5f")
    wio.save_weights_file(p, w, names)
    raw = open(p, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8:16] == bytes([0, 0, 0, 0, 0, 8, 8, 0])        # superblock v0, 8-byte offsets / lengths
    assert struct.unpack_from("<HH", raw, 16) == (4, 16)                                            # group leaf / internal K, as shipped
    base, free, eof, drv = struct.unpack_from("<QQQQ", raw, 24)
    assert base == 0 and eof == len(raw) and free == drv == 0xFFFFFFFFFFFFFFFF and len(raw) % 8 == 0
    back = wio.load_weights_file(p)
    assert all(np.array_equal(a, b) and b.dtype == np.float32 for a, b in zip(w, back))
    ds = h.read_datasets(p)
    assert sorted(ds) == sorted([f"/{n}/{n}/{t}:0" for n in names[:-1] for t in ("kernel", "bias")] +
                                ["/dense_3/dense_3_1/kernel:0", "/dense_3/dense_3_1/bias:0"])               # keras-rl's dueling layer scope
    at = h.read_attributes(p)
    assert at["/"] == {"layer_names": [n.encode() for n in names], "backend": b"tensorflow", "keras_version": b"2.2.2"}
    assert at["/conv2d_2"] == {"weight_names": [b"conv2d_2/kernel:0", b"conv2d_2/bias:0"]}
    assert at["/dense_3"] == {"weight_names": [b"dense_3_1/kernel:0", b"dense_3_1/bias:0"]}
    f = h._File(p)
    for addr in [f.root_header] + [c for _, c in f.children(f.root_header)]:
        kids = [n for n, _ in f.children(addr)]
        assert kids == sorted(kids) and addr % 8 == 0
    # a non-dueling network keeps the plain scope for its last layer
    p2 = str(tmp_path / "w.h5")
    wio.save_weights_file(p2, w[:10], names[:5], dueling=False)
    assert "/dense_2/dense_2/kernel:0" in h.read_datasets(p2)


def test_dropin_module_tree():
    """`from Environments import *`, `from rl.agents.dqn import DQNAgent` ... resolve to the GPU-backed classes."""
    sys.path.insert(0, os.path.join(ROOT, "deepq-decoding_amd", "dropin"))
    try:
        for m in ("Environments", "Function_Library", "rl", "rl.agents.dqn", "rl.policy", "rl.memory", "rl.callbacks"):
            sys.modules.pop(m, None)
        import Environments
        import Function_Library
        from rl.agents.dqn import DQNAgent
        from rl.callbacks import FileLogger
        from rl.memory import SequentialMemory
        from rl.policy import BoltzmannQPolicy, EpsGreedyQPolicy, GreedyQPolicy, LinearAnnealedPolicy
        for name in ("generateSurfaceCodeLattice", "multiplyPaulis", "generate_error", "generate_surface_code_syndrome_NoFT_efficient",
                     "generate_faulty_syndrome", "obtain_new_error_configuration", "index_to_move",
                     "generate_one_hot_labels_surface_code", "build_convolutional_nn"):
            assert callable(getattr(Function_Library, name)) and callable(getattr(Environments, name))
        assert Environments.Surface_Code_Environment_Multi_Decoding_Cycles.__name__ == "Surface_Code_Environment_Multi_Decoding_Cycles"
        assert all(callable(x) for x in (DQNAgent, FileLogger, SequentialMemory, EpsGreedyQPolicy, GreedyQPolicy, LinearAnnealedPolicy, BoltzmannQPolicy))
    finally:
        sys.path.pop(0)


def test_keras_and_gym_standins_recognise_the_reference_architecture():
    """The driver scripts build their network with keras.models.Sequential + layer objects (Single_Point_Training_Script.py:61-90): the
    stand-ins under dropin/ turn exactly that layer sequence into the model description, and refuse anything else loudly."""
    sys.path.insert(0, os.path.join(ROOT, "deepq-decoding_amd", "dropin"))
    try:
        for m in [k for k in sys.modules if k == "keras" or k.startswith("keras.") or k == "gym" or k.startswith("gym.")]:
            sys.modules.pop(m)
        import gym
        from keras.layers import Activation, Conv2D, Dense, Dropout, Flatten, MaxPooling2D
        from keras.layers.advanced_activations import LeakyReLU                     # noqa: F401  (imported by the scripts, unused)
        from keras.layers.normalization import BatchNormalization                   # noqa: F401
        from keras.models import Sequential, load_model
        from keras.optimizers import Adam
        from keras.utils import np_utils
        model = Sequential()
        model.add(Conv2D(filters=64, kernel_size=3, strides=2, input_shape=(7, 11, 11), data_format="channels_first"))
        model.add(Activation("relu"))
        for f, k, st in ((32, 2, 1), (32, 2, 1)):
            model.add(Conv2D(filters=f, kernel_size=k, strides=st, data_format="channels_first"))
            model.add(Activation("relu"))
        model.add(Flatten())
        model.add(Dense(512))
        model.add(Activation("relu"))
        model.add(Dropout(rate=0.2))
        model.add(Dense(51))
        model.add(Activation("linear"))
        assert model._describe() == ([[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], (7, 11, 11), 51)
        assert model.output_shape == (None, 51) and model.c_layers == [[64, 3, 2], [32, 2, 1], [32, 2, 1]]
        assert Adam(lr=1e-4).lr == 1e-4 and gym.spaces.Box(low=0, high=1, shape=(7, 11, 11)).shape == (7, 11, 11)
        assert gym.spaces.Discrete(51).n == 51 and np_utils.to_categorical([1, 0], 3).tolist() == [[0, 1, 0], [1, 0, 0]]
        bad = Sequential([Conv2D(8, 3, input_shape=(4, 7, 7), data_format="channels_first"), Activation("relu"), MaxPooling2D()])
        with pytest.raises(NotImplementedError):
            bad._describe()
        with pytest.raises(NotImplementedError):
            Conv2D(8, 3, data_format="channels_last")
        with pytest.raises(NotImplementedError):
            load_model("static_decoder")
    finally:
        sys.path.pop(0)


REF_H5 = "/root/reference/trained_models/d5_dp/0.007/final_dqn_weights.h5f"


@pytest.mark.skipif(not os.path.exists(REF_H5), reason="reference checkout not present (GPU box)")
def test_hdf5_reader_on_shipped_keras_weights():
    """The pure-Python HDF5 reader returns the shipped agent's tensors in Keras order with the shapes SURVEY.md §8a-D1
    lists; the committed fixture tests/golden/keras_weights_d5_dp_0.007.npz is exactly this file's content."""
    h = importlib.import_module("deepq-decoding_amd.hdf5_reader")
    w = h.read_keras_weights(REF_H5)
    assert [x.shape for x in w] == [(3, 3, 7, 64), (64,), (2, 2, 64, 32), (32,), (2, 2, 32, 32), (32,), (288, 512), (512,),
                                    (512, 51), (51,), (51, 52), (52,)]
    fx = load_golden("keras_weights_d5_dp_0.007")
    assert all(np.array_equal(fx[f"w{i}"], w[i]) for i in range(12))
    wio = importlib.import_module("deepq-decoding_amd.weights_io")
    assert all(np.array_equal(a, b) for a, b in zip(wio.load_weights_file(REF_H5), w))
    names = sorted(h.read_datasets(REF_H5))
    assert names[0] == "/conv2d_1/conv2d_1/bias:0" and names[-1] == "/dense_3/dense_3_1/kernel:0"
    at = h.read_attributes(REF_H5)                                   # the attribute parser the writer's test relies on, on a real file
    assert at["/"]["layer_names"][:2] == [b"conv2d_1_input", b"conv2d_1"] and at["/dense_3"] == {"weight_names": [b"dense_3_1/kernel:0", b"dense_3_1/bias:0"]}


@pytest.mark.skipif(not os.path.exists(REF_H5), reason="reference checkout not present (GPU box)")
def test_weight_fixtures_regenerate_and_all_shipped_agents_load():
    """tools/gen_weight_fixtures.py reproduces the committed fixtures array for array, and the pure-Python HDF5 reader loads every one
    of the 14 shipped agents (trained_models/d5_x/*, d5_dp/*) with the tensor shapes of SURVEY.md 8a-D1."""
    import glob
    import sys
    tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    sys.path.insert(0, tools)
    try:
        import gen_weight_fixtures as gw
    finally:
        sys.path.pop(0)
    for family, p in gw.AGENTS:
        fx = load_golden(gw.fixture_name(family, p))
        fresh = gw.build(family, p)
        assert sorted(fx.files) == sorted(fresh)
        for k in fresh:
            assert np.array_equal(fx[k], fresh[k]), (family, p, k)
    tails, fresh_tails = load_golden("training_history_tails"), gw.build_history_tails()
    assert all(np.array_equal(tails[k], fresh_tails[k]) for k in fresh_tails) and sorted(tails.files) == sorted(fresh_tails)
    for family, p in gw.HISTORIES:                         # the two from-scratch training records (training_history.json + their configs)
        fx = load_golden(gw.history_fixture_name(family, p))
        fresh = gw.build_history(family, p)
        assert sorted(fx.files) == sorted(fresh)
        for k in fresh:
            assert np.array_equal(fx[k], fresh[k], equal_nan=fresh[k].dtype.kind == "f"), (family, p, k)
    h = importlib.import_module("deepq-decoding_amd.hdf5_reader")
    files = sorted(glob.glob("/root/reference/trained_models/*/*/final_dqn_weights.h5f"))
    assert len(files) == 14
    for f in files:
        C, A = (6, 26) if "/d5_x/" in f else (7, 51)
        w = h.read_keras_weights(f)
        assert [x.shape for x in w] == [(3, 3, C, 64), (64,), (2, 2, 64, 32), (32,), (2, 2, 32, 32), (32,), (288, 512), (512,),
                                        (512, A), (A,), (A, A + 1), (A + 1,)], f
        assert all(x.dtype == np.float32 and np.isfinite(x).all() for x in w)


def test_grid_files_follow_the_reference_schema(tmp_path):
    """runner.write_grid reproduces the reference's config tree: the 36-point initial grid of
    Generate_Base_Configs_and_Simulation_Scripts.py:36-74 in its loop order, dict keys of :10-27 / :54-63; load_configs merges them like
    Single_Point_Training_Script.py:36-53; the Controller's selection rule (Controller.py:117-156) and spawning (:189-268)."""
    import pickle
    runner = importlib.import_module("deepq-decoding_amd.runner")
    fixed = {"d": 5, "use_Y": False, "train_freq": 1, "batch_size": 32, "print_freq": 250, "rolling_average_length": 1000,
             "stopping_patience": 1000, "error_model": "DP", "c_layers": [[64, 3, 2], [32, 2, 1], [32, 2, 1]], "ff_layers": [[512, 0.2]],
             "max_timesteps": 1000000, "volume_depth": 5, "testing_length": 101, "buffer_size": 50000, "dueling": True,
             "masked_greedy": False, "static_decoder": True}
    fam = str(tmp_path / "d5_dp")
    dirs = runner.write_grid(fam, fixed, 0.001, 100000)
    assert len(dirs) == 36 and os.path.basename(dirs[0]) == "config_1" and os.path.isdir(os.path.join(fam, "0.001", "output_files"))
    assert pickle.load(open(os.path.join(fam, "fixed_config.p"), "rb")) == fixed
    v1 = pickle.load(open(os.path.join(dirs[0], "variable_config_1.p"), "rb"))
    assert v1 == {"p_phys": 0.001, "p_meas": 0.001, "success_threshold": 100000, "learning_starts": 1000, "learning_rate": 0.0001,
                  "exploration_fraction": 100000, "max_eps": 1.0, "target_network_update_freq": 2500, "gamma": 0.99, "final_eps": 0.04}
    v2 = pickle.load(open(os.path.join(dirs[1], "variable_config_2.p"), "rb"))
    v4 = pickle.load(open(os.path.join(dirs[3], "variable_config_4.p"), "rb"))
    assert v2["final_eps"] == 0.02 and v4["target_network_update_freq"] == 5000 and v4["final_eps"] == 0.04     # innermost loops first
    v36 = pickle.load(open(os.path.join(dirs[35], "variable_config_36.p"), "rb"))
    assert (v36["learning_rate"], v36["exploration_fraction"], v36["final_eps"]) == (0.00001, 200000, 0.001)
    cfg, number = runner.load_configs(dirs[6])
    assert number == "7" and cfg["d"] == 5 and cfg["learning_rate"] == v1["learning_rate"] and set(cfg) == set(fixed) | set(v1)
    if os.path.exists("/root/reference/trained_models/d5_dp/fixed_config.p"):        # the reference's own pickles carry the same keys
        ref_fixed = pickle.load(open("/root/reference/trained_models/d5_dp/fixed_config.p", "rb"))
        assert set(ref_fixed) == set(runner.FIXED_KEYS) and ref_fixed == fixed
        ref_var = pickle.load(open("/root/reference/trained_models/d5_dp/0.011/variable_config_92.p", "rb"))
        assert set(ref_var) == set(runner.VARIABLE_KEYS)
    # Controller bookkeeping
    assert runner.collect_results(os.path.join(fam, "0.001"))["3"] == "not started"
    assert runner.select_best({"1": 900.0, "2": 1200.0, "3": "still running", "4": 1500.0}, 1000, 1) == {"4": 1500.0}
    assert runner.select_best({"1": 900.0}, 1000, 1) == {}
    for n, life in (("2", 1200.0), ("5", 1700.0)):
        cdir = os.path.join(fam, "0.001", f"config_{n}")
        pickle.dump([life / 2, life], open(os.path.join(cdir, "results.p"), "wb"))
        open(os.path.join(cdir, "final_dqn_weights.h5f"), "wb").write(b"weights" + n.encode())
        open(os.path.join(cdir, "memory.p"), "wb").write(b"memory" + n.encode())
    new = runner.spawn_next(fam, fixed, 0.001, 0.003)
    assert len(new) == 4 * 2 * 3 * 2 * 3                                          # Controller.py:27-33 grid from the ONE best point
    assert open(os.path.join(new[0], "initial_dqn_weights.h5f"), "rb").read() == b"weights5"
    assert open(os.path.join(new[-1], "memory.p"), "rb").read() == b"memory5"
    assert pickle.load(open(os.path.join(new[0], "variable_config_1.p"), "rb"))["p_phys"] == 0.003
    assert open(os.path.join(fam, "results", "best_results_from_0.001.txt")).read() == "5: 1700.0\n"
    assert open(os.path.join(fam, "results", "results_from_0.001.txt")).read().splitlines()[1] == "2: 1200.0"


def test_feed_forward_referee_from_a_keras_weight_file(tmp_path):
    """The reference's static decoder is a Keras feed-forward classifier loaded with load_model (TRAIN:54-57) and used through
    .predict (ENV:144).  FeedForwardReferee rebuilds a Dense stack from a Keras save_weights file (HDF5 written / read by the package's
    own codecs, or .npz) and answers .predict with softmax scores; the drop-in keras.models.load_model resolves to it."""
    ref_mod = importlib.import_module("deepq-decoding_amd.referee")
    wio = importlib.import_module("deepq-decoding_amd.weights_io")
    rng = np.random.RandomState(3)
    sizes = [36, 48, 20, 4]
    weights = []
    for a, b in zip(sizes, sizes[1:]):
        weights += [rng.randn(a, b).astype(np.float32) * 0.3, rng.randn(b).astype(np.float32) * 0.1]
    x = (rng.rand(50, 36) < 0.2).astype(np.int64)
    h = x.astype(np.float64)
    for i in range(0, len(weights), 2):
        h = h @ weights[i].astype(np.float64) + weights[i + 1]
        if i + 2 < len(weights):
            h = np.maximum(h, 0)
    want = np.exp(h - h.max(1, keepdims=True)); want /= want.sum(1, keepdims=True)
    for ext in ("h5f", "npz"):
        path = str(tmp_path / f"referee.{ext}")
        wio.save_weights_file(path, weights, ["dense_1", "dense_2", "dense_3"], dueling=False)
        r = ref_mod.FeedForwardReferee.from_file(path)
        assert (r.n_inputs, r.n_classes) == (36, 4)
        got = r.predict(x, batch_size=1, verbose=0)
        assert got.shape == (50, 4) and np.abs(got - want).max() < 1e-5 and np.allclose(got.sum(1), 1, atol=1e-5)
    # through the reference-named module tree
    dropin = os.path.join(ROOT, "deepq-decoding_amd", "dropin")
    code = ("import sys; sys.path.insert(0, %r); from keras.models import load_model; import numpy as np; "
            "m = load_model(%r); print(m.predict(np.zeros((1, 36)), batch_size=1, verbose=0).shape)") % (dropin, str(tmp_path / "referee.h5f"))
    import subprocess
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=str(tmp_path))
    assert out.returncode == 0 and "(1, 4)" in out.stdout, out.stderr


def test_feed_forward_referee_exact_restatement_of_the_device_arithmetic():
    """FeedForwardReferee.logits_exact / predict_exact restate the device referee's FIXED arithmetic (csrc/env.hip referee_mlp_kernel:
    float32, bias first, inputs in increasing index order, one rounded multiply and one rounded add per term, ReLU between layers, first
    maximum) -- checked here against an independent scalar loop in numpy float32, against the float64 / BLAS-ordered .predict wherever its top
    two outputs are not within round-off, and for the flat weight layout dq_env_set_referee_mlp takes (kernel (in, out) row-major, then
    the bias, per layer).  (The device side of the same equality: tests/test_env_gpu.py::test_dense_stack_referee_evaluated_on_the_device.)"""
    ref_mod = importlib.import_module("deepq-decoding_amd.referee")
    rng = np.random.RandomState(8)
    dims = [64, 40, 24, 4]
    weights = []
    for a, b in zip(dims, dims[1:]):
        weights += [(rng.randn(a, b) * (1.5 / np.sqrt(a))).astype(np.float32), (rng.randn(b) * 0.3).astype(np.float32)]
    r = ref_mod.FeedForwardReferee(weights)
    assert r.dims == dims
    flat = r.flat_weights()
    assert flat.dtype == np.float32 and flat.size == sum(a * b + b for a, b in zip(dims, dims[1:]))
    off = 0
    for (a, b), k, bias in zip(zip(dims, dims[1:]), weights[0::2], weights[1::2]):
        assert np.array_equal(flat[off:off + a * b].reshape(a, b), k) and np.array_equal(flat[off + a * b:off + a * b + b], bias)
        off += a * b + b
    x = (rng.rand(300, 64) < 0.15).astype(np.float32)
    z = r.logits_exact(x)
    assert z.dtype == np.float32 and z.shape == (300, 4)

    def scalar(row):                                                 # the kernel's loop, one output at a time
        h = row.astype(np.float32)
        for li, (k, bias) in enumerate(zip(weights[0::2], weights[1::2])):
            out = np.empty(k.shape[1], np.float32)
            for o in range(k.shape[1]):
                acc = np.float32(bias[o])
                for j in range(k.shape[0]):
                    acc = np.float32(acc + np.float32(h[j] * k[j, o]))
                out[o] = acc
            h = np.maximum(out, np.float32(0)) if li + 1 < len(weights) // 2 else out
        return h

    for i in range(0, 300, 37):
        assert np.array_equal(z[i], scalar(x[i])), i
    p = r.predict(x)
    top2 = np.sort(z, axis=1)[:, -2:]
    clear = top2[:, 1] - top2[:, 0] > 1e-4
    assert clear.mean() > 0.95 and np.array_equal(np.argmax(p, axis=1)[clear], np.argmax(z, axis=1)[clear])
    pe = r.predict_exact(x)
    assert np.array_equal(pe.sum(1), np.ones(300)) and np.array_equal(np.argmax(pe, axis=1), np.argmax(z, axis=1))


def test_replay_permutation_restatement_is_a_uniform_bijection():
    """oracle/memory_oracle.py replay_permute (the restatement of csrc/common.h dq_replay_permute, compared bit for bit with the device
    in tests/test_qnet_gpu.py::test_replay_sample_rule): a bijection of [0, M) for any M, different for different keys, and a fixed
    position is spread evenly over the range as the key changes."""
    from oracle import memory_oracle as M, philox
    for size in (1, 2, 3, 5, 136, 3008, 4096 * 254):
        keys = philox.philox4x32((7, 0, 0xFFFFFFFF, 0xFFFF | (philox.STREAM_REPLAY << 16)), (1, 2))
        n = min(size, 100000)
        out = M.replay_permute(np.arange(n), size, keys)
        assert out.max() < size and len(set(out.tolist())) == n
    size = 97
    hits = np.zeros(size, int)
    perms = set()
    for t in range(1, 2001):
        keys = philox.philox4x32((t, 0, 0xFFFFFFFF, 0xFFFF | (philox.STREAM_REPLAY << 16)), (1, 2))
        out = M.replay_permute(np.arange(size), size, keys)
        hits[out[5]] += 1
        perms.add(tuple(out[:6].tolist()))
    assert hits.min() > 5 and hits.max() < 45 and len(perms) > 1990
    # first draws of a minibatch: distinct rows; the whole candidate set when batch == M
    z = np.zeros((20, 8), np.uint8)
    rows = M.device_replay_rows(z, 8, 20, 5, 20, 136, (8, 9), 3)
    assert set(rows.tolist()) == M.valid_transitions(z, 8, 20, 5, 20)


def test_live_timing_samples_the_launches():
    """bench.py times a SAMPLE of the dominant kernel's launches inside the timed region (a timed launch costs its stream a few microseconds):
    about 64 over a long region, every fourth launch of a 20-step run, every launch of a very short one."""
    import importlib
    bl = importlib.import_module("deepq-decoding_amd.bench_loop")
    assert bl.prof_stride(2000) == 31 and 2000 // bl.prof_stride(2000) >= 64
    assert bl.prof_stride(256) == 4
    assert bl.prof_stride(20) == 4 and 20 // bl.prof_stride(20) == 5
    assert bl.prof_stride(5) == 1 and bl.prof_stride(1) == 1


def test_pmc_stamp_matches_sources():
    """`roofline.traffic` of the driver's line comes from the committed PMC passes (profiles/pmc_traffic_*.json), which bench.py refuses when they were
    taken with other kernel code (round 5: a comment edit orphaned them and BENCH_r05 carried `traffic: null`).  (a) the stamp hashes CODE only -- comments
    and whitespace do not move it; (b) the committed passes carry the digest of the sources at HEAD and answer for the symbols the default runs launch."""
    import importlib
    bl = importlib.import_module("deepq-decoding_amd.bench_loop")
    a = "int f(int x) { /* doc */ return x + 1;   // trailing\n}\n"
    b = "int f(int x){\n  return x+1;\n}   /* other\n comment */\n"
    assert bl._code_only(a) == bl._code_only(b) and bl._code_only(a) != bl._code_only(a.replace("+ 1", "+ 2"))
    digest = bl.csrc_digest()
    for mode, config, symbol in (("loop", "c3", "conv_wave_kernel"), ("loop", "c3", "conv_bwd16_kernel"), ("loop", "c3", "dense_chain_kernel"),
                                 ("loop", "c5", "dense_chain_kernel"), ("loop", "c2", "conv_wave_kernel"), ("env", "c3", "env_multi_kernel")):
        rec = bl.pmc_record(mode, config)
        assert rec is not None, f"profiles/pmc_traffic_{mode}_{config}.json is missing"
        assert rec["csrc_sha256"] == digest, (f"profiles/pmc_traffic_{mode}_{config}.json was measured with other kernel code ({rec['csrc_sha256'][:12]} != "
                                              f"{digest[:12]}): run tools/restamp_pmc.sh on the GPU box and commit profiles/pmc_traffic_*.json")
        got = bl.pmc_traffic(symbol, mode, config)
        assert got is not None and got > 0, (mode, config, symbol)
    # a family alias or a form that did not run is NOT answered with another kernel's bytes
    assert bl.pmc_traffic("conv_chain_kernel", "loop", "c3") is None
    assert bl.pmc_traffic("conv_wave_kernel", "loop", "c3", minibatch=32) is None


@pytest.mark.parametrize("name", ["training_history_d5_x_0.001", "training_history_d5_dp_0.001"])
def test_reference_mean_eps_records_pin_the_step_arithmetic(dq, name):
    """The reference's own training records (trained_models/<family>/0.001/training_history.json, committed as arrays) pin keras-rl's step
    arithmetic for the fork, which is not in the tree: every recorded `mean_eps` is the mean of LinearAnnealedPolicy's epsilon
    (value_max - (value_max - value_min) s / nb_steps, floored at value_min) over exactly the episode's steps s with s > learning_starts --
    i.e. epsilon is read with the 0-based step number BEFORE Agent.fit increments it, metrics are NaN on steps without an update (nan-mean
    per episode), and the first update happens at s = learning_starts + 1.  DQNAgent's host logic is then checked against that rule."""
    import json
    g = load_golden(name)
    var = json.loads(str(g["variable_config_json"]))
    S, L, me = g["nb_steps"], g["nb_episode_steps"], g["mean_eps"]
    assert np.array_equal(np.diff(np.concatenate([[0], S])), L)            # nb_steps is cumulative: episode i took steps [S[i-1], S[i])
    warm, n_anneal, hi, lo = var["learning_starts"], var["exploration_fraction"], var["max_eps"], var["final_eps"]
    eps = lambda s: np.maximum(lo, hi - (hi - lo) * s / float(n_anneal))
    first = int(np.argmax(~np.isnan(me)))
    assert S[first - 1] <= warm + 1 < S[first]                              # the first episode with a trained step is the one holding s = warm + 1
    assert np.isnan(g["loss"][:first]).all() and np.isnan(g["mean_q"][:first]).all() and np.isnan(me[:first]).all()
    assert not np.isnan(g["loss"][first:]).any()
    worst = 0.0
    for i in list(range(first, min(first + 2000, len(S)))) + list(range(len(S) - 50, len(S))):
        s = np.arange(S[i - 1], S[i])
        s = s[s > warm]
        worst = max(worst, abs(float(eps(s).mean()) - float(me[i])))
    assert worst < 1e-12, worst
    # an alternative arithmetic (epsilon read after the increment, or the first update at s = warm) misses the records by ~5e-6 / 2.5e-6
    i = first
    assert abs(float(eps(np.arange(S[i - 1], S[i])[np.arange(S[i - 1], S[i]) >= warm]).mean()) - float(me[i])) > 1e-7
    # DQNAgent's host-side rule (agent.py _will_train / _sync_target), on a stand-in for the device loop
    agent = dq.DQNAgent.__new__(dq.DQNAgent)
    agent.nb_steps_warmup, agent.train_interval, agent.target_model_update, agent._last_target_sync = warm, 1, var["target_network_update_freq"], 0

    class _Core:
        N, T, filled = 1, 50001, 50001
        copies = []

        def update_target_hard(self):
            self.copies.append(agent.step - 1)
    agent._core = _Core()
    trained = [s for s in range(0, warm + 5) if agent._will_train(s + 1)]
    assert trained == [warm + 1, warm + 2, warm + 3, warm + 4]
    for s in range(0, 2 * agent.target_model_update + 3):
        agent.step = s + 1
        agent._sync_target()
    assert agent._core.copies == [0, agent.target_model_update, 2 * agent.target_model_update]      # keras-rl: step % target_model_update == 0


def test_reference_records_pin_the_early_stopping_rule():
    """All fourteen shipped training runs (their early-stopping bookkeeping, committed as tests/golden/training_history_tails.npz): a run whose
    time_since_best passed stopping_patience BEFORE min_nb_steps carried on (up to 9426 episodes without a new best in d5_dp/0.001); past
    min_nb_steps every run tolerated a count equal to the patience and ended on the first episode that exceeded it, that row being the only one
    with stopped_improving set; the others ran to max_timesteps.  agent.stopping_flags restates exactly that."""
    import json
    agent = importlib.import_module("deepq-decoding_amd.agent")
    t = load_golden("training_history_tails")
    rows = json.loads(str(t["rows_json"]))
    assert len(rows) == 14
    stopped = 0
    for name, r in zip(t["agents"], rows):
        pat, mn = r["patience"], r["min_nb_steps"]
        assert not r["any_stopped_before_last"], name
        assert r["max_tsb_after_min_before_last"] <= pat, name                  # never more than the patience once min_nb_steps had passed ...
        for steps, tsb, flag in zip(r["tail_nb_steps"], r["tail_time_since_best"], r["tail_stopped_improving"]):
            _, mine = agent.stopping_flags(False, False, 0.0, tsb, steps, r["success_threshold"], pat, mn)
            assert mine == flag, (name, steps, tsb)
        if r["tail_stopped_improving"][-1]:
            stopped += 1
            assert r["tail_time_since_best"][-1] == pat + 1 and r["tail_nb_steps"][-1] >= mn and r["tail_nb_steps"][-1] < r["max_timesteps"] - 2000
        else:
            assert r["tail_nb_steps"][-1] > r["max_timesteps"] - 2000, name     # ... or the run used up its steps
        # before min_nb_steps the count runs past the patience without effect
        assert agent.stopping_flags(False, False, 0.0, r["max_tsb_before_min"], mn - 1, r["success_threshold"], pat, mn) == (False, False)
    assert stopped == 9 and max(r["max_tsb_before_min"] for r in rows) > 9000
