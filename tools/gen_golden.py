#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference); the GPU box never sees
the reference, only the arrays this script writes.  Usage:

    python tools/gen_golden.py            # writes tests/golden/*.npz
    python tools/gen_golden.py --time     # also times the reference env (prints JSON)

What it does
  * imports /root/reference/cluster_scripts/d5_dp/Function_Library.py and
    /root/reference/example_notebooks/Environments.py UNMODIFIED, with a minimal
    in-memory ``gym`` stub (the reference only touches gym.spaces.Box/Discrete,
    Environments.py:78-84);
  * replaces ``np.random.rand`` / ``np.random.randint`` (looked up through the module
    at call time, Function_Library.py:99-100,191) with a replay object that serves
    the site-indexed Philox words of oracle/philox.py in the reference's own call
    order;
  * installs the build's look-up referee (oracle/referee.py) as ``static_decoder``
    (the reference's own referee blobs are missing, .MISSING_LARGE_BLOBS:1-4);
  * records tables, known-answer vectors and episode traces.

No reference source text is written anywhere: fixtures hold inputs and outputs only.
"""
import argparse
import hashlib
import json
import os
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")

from oracle import philox, lattice, referee  # noqa: E402

SEED = (0x5EED, 0xD0DEC0DE)


# ---------------------------------------------------------------------------------------------
def import_reference():
    gym = types.ModuleType("gym")
    spaces = types.ModuleType("gym.spaces")

    class Box:
        def __init__(self, low, high, shape, dtype):
            self.low, self.high, self.shape, self.dtype = low, high, tuple(shape), dtype

    class Discrete:
        def __init__(self, n):
            self.n = n

    spaces.Box, spaces.Discrete = Box, Discrete
    gym.spaces = spaces
    sys.modules["gym"] = gym
    sys.modules["gym.spaces"] = spaces
    sys.path.insert(0, os.path.join(REF, "cluster_scripts", "d5_dp"))   # keras-free Function_Library
    import Function_Library as FL                                       # noqa: E402
    sys.path.insert(0, os.path.join(REF, "example_notebooks"))
    import importlib.util
    spec = importlib.util.spec_from_file_location("Environments", os.path.join(REF, "example_notebooks", "Environments.py"))
    ENV = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ENV)
    return FL, ENV


class SiteStream:
    """Serves np.random.rand()/randint(1,4) from the site-indexed Philox stream, in the reference's
    call order: per measurement round d*d `rand` (row-major qubits, each hit followed by exactly
    one `randint`), then d*d-1 `rand` in the generate_faulty_syndrome order."""

    def __init__(self, d, seed, env_id):
        self.d2 = d * d
        self.seed, self.env_id = seed, env_id
        self.round = 0
        self.pos = 0
        self.last_hit = False

    def rand(self):
        if self.pos < self.d2:
            w = philox.site_words(self.seed, self.env_id, self.round, self.pos)[0]
        else:
            w = philox.site_words(self.seed, self.env_id, self.round, self.pos - self.d2)[2]
        self.pos += 1
        if self.pos == 2 * self.d2 - 1:
            self.pos = 0
            self.round += 1
        return w / 4294967296.0

    def randint(self, lo, hi):
        assert (lo, hi) == (1, 4) and 1 <= self.pos <= self.d2
        return philox.pauli_type(philox.site_words(self.seed, self.env_id, self.round, self.pos - 1)[1])

    def at_round_boundary(self):
        return self.pos == 0


class SiteStreamIIDXZ(SiteStream):
    """generate_IIDXZ_error (FL:134-160) draws TWO uniforms per qubit (X flip, then Z flip) and no integer: per round 2 d*d `rand`
    (qubit q: site words k0, k1), then d*d-1 `rand` in the generate_faulty_syndrome order (word k2)."""

    def rand(self):
        n_err = 2 * self.d2
        if self.pos < n_err:
            w = philox.site_words(self.seed, self.env_id, self.round, self.pos // 2)[self.pos % 2]
        else:
            w = philox.site_words(self.seed, self.env_id, self.round, self.pos - n_err)[2]
        self.pos += 1
        if self.pos == n_err + self.d2 - 1:
            self.pos = 0
            self.round += 1
        return w / 4294967296.0

    def randint(self, lo, hi):
        raise AssertionError("the IIDXZ channel draws no integers")


class ListStream:
    def __init__(self, words):
        self.words, self.i = list(words), 0

    def rand(self):
        w = self.words[self.i]
        self.i += 1
        return int(w) / 4294967296.0


def policy_action(seed, env_id, t, legal_set, n_actions):
    """Golden-trace policy: 7/8 uniform over the legal set (k-th smallest), 1/8 uniform over ALL
    actions (exercises repeated and non-legal moves).  Words from STREAM_POLICY."""
    w = philox.site_words(seed, env_id, t, 0, stream=philox.STREAM_POLICY)
    if (w[1] >> 29) == 0:
        return philox.bounded(w[2], n_actions)
    legal = sorted(legal_set)
    return legal[philox.bounded(w[0], len(legal))]


def set_to_mask2(s):
    lo = sum(1 << a for a in s if a < 64)
    hi = sum(1 << (a - 64) for a in s if a >= 64)
    return np.array([lo, hi], dtype=np.uint64)


def set_to_words(s, n_words):
    v = sum(1 << a for a in s)
    return np.array([(v >> (64 * k)) & 0xFFFFFFFFFFFFFFFF for k in range(n_words)], dtype=np.uint64)


# ---------------------------------------------------------------------------------------------
def gen_tables(FL, ENV, out):
    for d in (3, 5, 7):
        env = ENV.Surface_Code_Environment_Multi_Decoding_Cycles(d=d, error_model="DP", use_Y=False, volume_depth=d)
        out[f"qubits_d{d}"] = env.qubits.astype(np.int16)
        stabs = np.full((d * d, 4, 2), -1, dtype=np.int16)
        for q, lst in enumerate(env.qubit_stabilizers):
            for k, ab in enumerate(lst):
                stabs[q, k] = ab
        out[f"qubit_stabilizers_d{d}"] = stabs
        neigh = np.full((d * d, 8), -1, dtype=np.int16)
        for q, lst in enumerate(env.qubit_neighbours):
            neigh[q, :len(lst)] = lst
        out[f"qubit_neighbours_d{d}"] = neigh
        out[f"identity_indicator_d{d}"] = env.identity_indicator.astype(np.uint8)
        out[f"static_plane_d{d}"] = env.padding_syndrome(np.zeros((d + 1, d + 1), int)).astype(np.uint8)
        for model, use_Y in (("X", False), ("DP", True), ("DP", False)):
            e = ENV.Surface_Code_Environment_Multi_Decoding_Cycles(d=d, error_model=model, use_Y=use_Y, volume_depth=d)
            out[f"meta_d{d}_{model}_{int(use_Y)}"] = np.array(
                [e.num_actions, e.n_action_layers, e.identity_index, *e.observation_space.shape], dtype=np.int32)


def gen_kats(FL, ENV, out):
    rng = np.random.RandomState(12345)      # only chooses INPUTS; isolated from the patched global stream
    for d in (3, 5, 7):
        qubits = FL.generateSurfaceCodeLattice(d)
        errs = []
        for q in range(d * d):              # G2a: every single-qubit X/Y/Z error
            for p in (1, 2, 3):
                e = np.zeros((d, d), int)
                e[q // d, q % d] = p
                errs.append(e)
        for _ in range(1000):               # G2b: random Pauli configurations
            dens = rng.choice([0.05, 0.2, 0.75])
            e = (rng.rand(d, d) < dens) * rng.randint(1, 4, size=(d, d))
            errs.append(e.astype(int))
        errs = np.array(errs)
        out[f"kat_err_d{d}"] = errs.astype(np.uint8)
        out[f"kat_syn_d{d}"] = np.array([FL.generate_surface_code_syndrome_NoFT_efficient(e, qubits) for e in errs]).astype(np.uint8)
        out[f"kat_label_dp_d{d}"] = np.array([FL.generate_one_hot_labels_surface_code(e, "DP") for e in errs]).astype(np.uint8)
        out[f"kat_label_x_d{d}"] = np.array([FL.generate_one_hot_labels_surface_code(e * (e == 1), "X") for e in errs]).astype(np.uint8)
        # multiplyPaulis table + obtain_new_error_configuration on random pairs (E2, E8)
        out[f"kat_mul_d{d}"] = np.array([[FL.multiplyPaulis(a, b) for b in range(4)] for a in range(4)], dtype=np.uint8)
        a, b = errs[-200:-100], errs[-100:]
        out[f"kat_prod_d{d}"] = np.array([FL.obtain_new_error_configuration(x, y) for x, y in zip(a, b)]).astype(np.uint8)
        # G3: faulty syndrome under injected words (pins the draw order of FL:189-221)
        n_stab = d * d - 1
        trues, words, faults = [], [], []
        saved = np.random.rand
        for i in range(64):
            t = out[f"kat_syn_d{d}"][3 * d * d + i].astype(int)
            if i < n_stab:                  # flip exactly the i-th draw
                w = np.full(n_stab, 0xFFFFFFFF, dtype=np.uint64)
                w[i] = 0
                p = 0.5
            else:
                w = rng.randint(0, 2 ** 32, size=n_stab, dtype=np.uint64)
                p = 0.3
            stream = ListStream(w)
            np.random.rand = stream.rand
            f = FL.generate_faulty_syndrome(t, p)
            np.random.rand = saved
            assert stream.i == n_stab
            trues.append(t), words.append(w), faults.append(f)
        out[f"kat_faulty_true_d{d}"] = np.array(trues, dtype=np.uint8)
        out[f"kat_faulty_words_d{d}"] = np.array(words, dtype=np.uint32)
        out[f"kat_faulty_p_d{d}"] = np.array([0.5] * n_stab + [0.3] * (64 - n_stab))
        out[f"kat_faulty_out_d{d}"] = np.array(faults, dtype=np.uint8)
        # G4: padding
        env = ENV.Surface_Code_Environment_Multi_Decoding_Cycles(d=d, error_model="DP", use_Y=False, volume_depth=d)
        syn_in = (rng.rand(32, d + 1, d + 1) < 0.3).astype(int)
        out[f"kat_padsyn_in_d{d}"] = syn_in.astype(np.uint8)
        out[f"kat_padsyn_out_d{d}"] = np.array([env.padding_syndrome(s) for s in syn_in]).astype(np.uint8)
        act_in = (rng.rand(32, d * d) < 0.3).astype(int)
        out[f"kat_padact_in_d{d}"] = act_in.astype(np.uint8)
        out[f"kat_padact_out_d{d}"] = np.array([env.padding_actions(a) for a in act_in]).astype(np.uint8)
        # G5: index_to_move for every action (and one past the identity)
        for model, use_Y in (("X", False), ("DP", True), ("DP", False)):
            n_act = lattice.num_actions(d, model, use_Y)[0]
            out[f"kat_move_d{d}_{model}_{int(use_Y)}"] = np.array(
                [FL.index_to_move(d, a, model, use_Y) for a in range(n_act + 1)]).astype(np.uint8)
    # G9: generate_error(d, p, "IIDXZ") under injected words (pins: two draws per qubit, X first; both hits -> Y)
    saved = np.random.rand
    for d in (3, 5, 7):
        words, errs_out = [], []
        for i in range(24):
            w = rng.randint(0, 2 ** 32, size=2 * d * d, dtype=np.uint64)
            stream = ListStream(w)
            np.random.rand = stream.rand
            e = FL.generate_error(d, 0.3, "IIDXZ")
            np.random.rand = saved
            assert stream.i == 2 * d * d
            words.append(w), errs_out.append(e)
        out[f"kat_iidxz_words_d{d}"] = np.array(words, dtype=np.uint32)
        out[f"kat_iidxz_err_d{d}"] = np.array(errs_out, dtype=np.uint8)
    # G7: the README known-answer vector (README.md:712-780): X on qubit (4,1) of d=5
    qubits = FL.generateSurfaceCodeLattice(5)
    e = np.zeros((5, 5), int)
    e[4, 1] = 1
    out["kat_readme_syn"] = FL.generate_surface_code_syndrome_NoFT_efficient(e, qubits).astype(np.uint8)
    # G8: logical operators at d=5
    g8_err, g8_syn, g8_lab = [], [], []
    for spec in ("row0_x", "col0_z", "col0_x", "col3_x", "col0_y"):
        e = np.zeros((5, 5), int)
        if spec == "row0_x":
            e[0, :] = 1
        elif spec == "col0_z":
            e[:, 0] = 3
        elif spec == "col0_x":
            e[:, 0] = 1
        elif spec == "col3_x":
            e[:, 3] = 1
        else:
            e[:, 0] = 2
        g8_err.append(e)
        g8_syn.append(FL.generate_surface_code_syndrome_NoFT_efficient(e, qubits))
        g8_lab.append(FL.generate_one_hot_labels_surface_code(e, "DP"))
    out["kat_logical_err"] = np.array(g8_err, dtype=np.uint8)
    out["kat_logical_syn"] = np.array(g8_syn, dtype=np.uint8)
    out["kat_logical_label"] = np.array(g8_lab, dtype=np.uint8)


TRACE_CONFIGS = {
    # name: (d, model, use_Y, p_phys, p_meas, depth, n_envs, n_steps)
    "c1_d3_x": (3, "X", False, 0.005, 0.005, 3, 16, 128),
    "c2_d5_x": (5, "X", False, 0.007, 0.007, 5, 16, 128),
    "c3_d5_dp": (5, "DP", False, 0.011, 0.011, 5, 16, 128),
    "c5_d7_dp": (7, "DP", False, 0.005, 0.005, 7, 16, 128),
    "x1_d5_dpy": (5, "DP", True, 0.03, 0.02, 3, 8, 96),        # Y actions, ctor-default depth
    "x2_d5_dp_hot": (5, "DP", False, 0.08, 0.08, 5, 8, 96),    # many failures / resets
    "x3_d3_x_nomeas": (3, "X", False, 0.02, 0.0, 3, 8, 96),    # p_meas = 0: long rejection loops
    "x4_d7_x": (7, "X", False, 0.01, 0.01, 4, 4, 64),          # depth != d
    # IIDXZ noise (generate_error(d, p, "IIDXZ"), FL:91-92,134-160).  The reference's environment constructor cannot be built with this
    # model string (ENV:66-69), so the trace runs its "DP" environment (same action layers, same four homology classes, FL:329-330) with
    # the name `generate_error` in the Environments module bound to the reference's own IIDXZ generator -- no reference code is changed.
    "x5_d5_iidxz": (5, "IIDXZ", False, 0.02, 0.015, 5, 8, 96),
    "x6_d7_iidxz": (7, "IIDXZ", False, 0.008, 0.008, 5, 4, 64),    # the same at d = 7 (table referee in L2, depth != d)
    "x7_d3_dpy": (3, "DP", True, 0.02, 0.02, 3, 8, 96),            # Y moves on the smallest lattice (28 actions)
}


# Lattices beyond one 64-bit word per bit-plane (the wide environment, csrc/env_big.hip): no look-up referee exists, the reference runs
# with oracle/matching_referee.py (same definition, pinned against the tables at d <= 7) as its static_decoder.  `legal` / `acted` are
# stored as little-endian 64-bit word arrays.
BIG_TRACE_CONFIGS = {
    "b1_d9_dp": (9, "DP", False, 0.004, 0.004, 9, 4, 64),       # planes of 2 words, 163 actions
    "b2_d9_x": (9, "X", False, 0.008, 0.008, 4, 4, 64),
    "b3_d11_dp": (11, "DP", False, 0.004, 0.003, 3, 3, 48),     # 121 qubits, 243 actions
    "b4_d13_x": (13, "X", False, 0.006, 0.006, 3, 2, 40),       # planes of 3 words
    "b5_d15_dpy": (15, "DP", True, 0.003, 0.003, 2, 2, 32),     # planes of 4 words, 676 actions
    "b6_d9_iidxz": (9, "IIDXZ", False, 0.006, 0.005, 4, 4, 64),  # independent X and Z flips on the wide environment (the reference's generator, as x5)
}


def run_trace(ENV, cfg, luts, auto_reset=True):
    d, model, use_Y, p_phys, p_meas, depth, n_envs, n_steps = cfg
    iidxz = model == "IIDXZ"
    env_model = "DP" if iidxz else model
    saved_gen = ENV.generate_error
    if iidxz:
        import Function_Library as FL
        ENV.generate_error = lambda d_, p_, m_: FL.generate_error(d_, p_, "IIDXZ")
    big = d > 7
    if big:
        from oracle import matching_referee
        ref = matching_referee.MatchingReferee(d, model)
    else:
        ref = referee.LutReferee(d, model, lut_x=luts[d][0], lut_z=luts[d][1])
    n_act = lattice.num_actions(d, model, use_Y)[0]
    LW, W = ((n_act + 63) // 64, (d * d + 63) // 64) if big else (2, 1)
    C, n = depth + lattice.num_actions(d, model, use_Y)[1], 2 * d + 1
    rec = dict(
        obs=np.zeros((n_envs, n_steps + 1, C, n, n), np.uint8),
        action=np.zeros((n_envs, n_steps), np.int32),
        reward=np.zeros((n_envs, n_steps), np.float32),
        done=np.zeros((n_envs, n_steps + 1), np.uint8),
        was_reset=np.zeros((n_envs, n_steps), np.uint8),
        lifetime=np.zeros((n_envs, n_steps + 1), np.int32),
        hidden=np.zeros((n_envs, n_steps + 1, d, d), np.uint8),
        true_syndrome=np.zeros((n_envs, n_steps + 1, d + 1, d + 1), np.uint8),
        summed_nonzero=np.zeros((n_envs, n_steps + 1, d + 1, d + 1), np.uint8),
        legal=np.zeros((n_envs, n_steps + 1, LW), np.uint64),
        completed=np.zeros((n_envs, n_steps + 1, n_act), np.uint8),
        acted=np.zeros((n_envs, n_steps + 1, W) if big else (n_envs, n_steps + 1), np.uint64),
        rounds=np.zeros((n_envs, n_steps + 1), np.int64),
    )
    saved = (np.random.rand, np.random.randint)
    try:
        for e in range(n_envs):
            stream = (SiteStreamIIDXZ if iidxz else SiteStream)(d, SEED, e)
            np.random.rand, np.random.randint = stream.rand, stream.randint
            env = ENV.Surface_Code_Environment_Multi_Decoding_Cycles(
                d=d, p_phys=p_phys, p_meas=p_meas, error_model=env_model, use_Y=use_Y, volume_depth=depth, static_decoder=ref)

            def snap(t):
                rec["obs"][e, t] = env.board_state
                rec["done"][e, t] = env.done
                rec["lifetime"][e, t] = env.lifetime
                rec["hidden"][e, t] = env.hidden_state
                rec["true_syndrome"][e, t] = env.current_true_syndrome
                rec["summed_nonzero"][e, t] = env.summed_syndrome_volume != 0
                rec["legal"][e, t] = set_to_words(env.legal_actions, LW) if big else set_to_mask2(env.legal_actions)
                rec["completed"][e, t] = env.completed_actions
                rec["acted"][e, t] = set_to_words(env.acted_on_qubits, W) if big else sum(1 << q for q in env.acted_on_qubits)
                rec["rounds"][e, t] = stream.round
                assert stream.at_round_boundary()

            obs = env.reset()
            assert obs is env.board_state
            snap(0)
            for t in range(n_steps):
                a = policy_action(SEED, e, t, env.legal_actions, n_act)
                rec["action"][e, t] = a
                if auto_reset and env.done:
                    # keras-rl style: the step after a terminal one is spent resetting; its action is ignored
                    env.reset()
                    rec["was_reset"][e, t] = 1
                    rec["reward"][e, t] = 0.0
                else:
                    obs, r, done, info = env.step(a)
                    assert obs is env.board_state and info == {}
                    rec["reward"][e, t] = r
                snap(t + 1)
    finally:
        np.random.rand, np.random.randint = saved
        ENV.generate_error = saved_gen
    rec["config"] = np.array([d, {"X": 0, "DP": 1, "IIDXZ": 2}[model], int(use_Y), depth, n_envs, n_steps], dtype=np.int32)
    rec["rates"] = np.array([p_phys, p_meas], dtype=np.float64)
    rec["seed"] = np.array(SEED, dtype=np.uint32)
    return rec


def time_reference(ENV, luts, seconds=5.0):
    """Reference env steps/s in this container (1 process), LUT referee, uniform-over-legal policy."""
    res = {}
    for name in ("c1_d3_x", "c2_d5_x", "c3_d5_dp", "c5_d7_dp"):
        d, model, use_Y, p_phys, p_meas, depth, _, _ = TRACE_CONFIGS[name]
        ref = referee.LutReferee(d, model, lut_x=luts[d][0], lut_z=luts[d][1])
        env = ENV.Surface_Code_Environment_Multi_Decoding_Cycles(
            d=d, p_phys=p_phys, p_meas=p_meas, error_model=model, use_Y=use_Y, volume_depth=depth, static_decoder=ref)
        rng = np.random.RandomState(1)
        env.reset()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            if env.done:
                env.reset()
            legal = sorted(env.legal_actions)
            env.step(legal[rng.randint(len(legal))])
            n += 1
        res[name] = n / (time.perf_counter() - t0)
    return res


_TIME_ARGS = None


def _time_worker(seconds):
    return time_reference(_TIME_ARGS[0], _TIME_ARGS[1], seconds)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--big-only", action="store_true", help="only the traces of the lattices beyond d = 7 (BIG_TRACE_CONFIGS)")
    ap.add_argument("--time-only", type=int, default=0, metavar="N",
                    help="no fixtures: only time the reference environment on N processes at once (independent lattices, steps summed: SURVEY.md 8d (1))")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    FL, ENV = import_reference()
    if args.time_only:
        import multiprocessing as mp
        luts = {d: (referee.build_lut(d, 3), referee.build_lut(d, 1)) for d in (3, 5, 7)}
        global _TIME_ARGS                                           # (forked workers inherit the imported reference module: it does not pickle)
        _TIME_ARGS = (ENV, luts)
        ctx = mp.get_context("fork")
        with ctx.Pool(args.time_only) as pool:
            per = pool.map(_time_worker, [8.0] * args.time_only)
        total = {k: sum(r[k] for r in per) for k in per[0]}
        print(json.dumps({"processes": args.time_only, "cores": os.cpu_count(), "reference_env_steps_per_s_total": total,
                          "per_process_min": {k: min(r[k] for r in per) for k in per[0]}, "per_process_max": {k: max(r[k] for r in per) for k in per[0]}}))
        return

    def big_traces():
        for name, cfg in BIG_TRACE_CONFIGS.items():
            t0 = time.time()
            rec = run_trace(ENV, cfg, None, auto_reset=True)
            np.savez_compressed(os.path.join(OUT, f"trace_{name}.npz"), **rec)
            print(f"trace {name}: {time.time() - t0:.1f}s  resets={int(rec['was_reset'].sum())} rewards={int((rec['reward'] > 0).sum())} "
                  f"done-steps={int(rec['done'].sum())}")

    if args.big_only:
        big_traces()
        return

    t0 = time.time()
    luts = {d: (referee.build_lut(d, 3), referee.build_lut(d, 1)) for d in (3, 5, 7)}
    print(f"referee LUTs built in {time.time() - t0:.1f}s")
    lut_out = {}
    for d in (3, 5):
        lut_out[f"lut_x_d{d}"], lut_out[f"lut_z_d{d}"] = luts[d]
    for d in (3, 5, 7):
        for nm, arr in zip("xz", luts[d]):
            lut_out[f"lut_{nm}_sha256_d{d}"] = np.frombuffer(hashlib.sha256(arr.tobytes()).digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(OUT, "referee_lut.npz"), **lut_out)

    tables = {}
    gen_tables(FL, ENV, tables)
    np.savez_compressed(os.path.join(OUT, "tables.npz"), **tables)

    kats = {}
    gen_kats(FL, ENV, kats)
    np.savez_compressed(os.path.join(OUT, "kats.npz"), **kats)

    for name, cfg in TRACE_CONFIGS.items():
        t0 = time.time()
        rec = run_trace(ENV, cfg, luts, auto_reset=True)
        np.savez_compressed(os.path.join(OUT, f"trace_{name}.npz"), **rec)
        print(f"trace {name}: {time.time() - t0:.1f}s  resets={int(rec['was_reset'].sum())} "
              f"identity-ish={(rec['action'] == lattice.num_actions(cfg[0], cfg[1], cfg[2])[0] - 1).mean():.3f}")
    # sticky-done traces: keep stepping after done without reset (ENV:151 never clears `done`)
    for name in ("c3_d5_dp", "x2_d5_dp_hot"):
        cfg = list(TRACE_CONFIGS[name])
        cfg[6], cfg[7] = 4, 48
        rec = run_trace(ENV, tuple(cfg), luts, auto_reset=False)
        np.savez_compressed(os.path.join(OUT, f"trace_sticky_{name}.npz"), **rec)

    big_traces()

    if args.time:
        print(json.dumps({"reference_env_steps_per_s_1proc": time_reference(ENV, luts), "cores": os.cpu_count()}))


if __name__ == "__main__":
    main()
