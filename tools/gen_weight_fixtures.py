#!/usr/bin/env python3
"""Generates the Keras-weight fixtures under tests/golden/ from the reference's shipped agents.

    python tools/gen_weight_fixtures.py            (build container only: reads /root/reference/trained_models)

For each agent listed in AGENTS: the twelve tensors of trained_models/<family>/<p>/final_dqn_weights.h5f in Keras order
(read with the package's own pure-Python HDF5 reader -- h5py is not installed), plus the lifetimes the reference recorded
for that agent (all_results.p: test error rate -> mean lifetime over testing_length = 101 episodes) and the agent's
variable_config (hyper-parameters).  The fixtures are DATA (tensors and numbers); no reference source text is stored.
tests/test_host_logic.py::test_weight_fixtures_regenerate re-reads the .h5f files here and checks the committed arrays.
"""
import glob
import importlib
import os
import pickle
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference/trained_models"
# (family, training error rate): every agent the reference ships (trained_models/d5_x/0.001 .. 0.015, d5_dp/0.001 .. 0.011): 14 weight sets,
# each with the whole test-rate sweep the reference recorded for it (all_results.p).  Rounds 1-2 committed three of them.
AGENTS = [(fam, p) for fam in ("d5_x", "d5_dp") for p in sorted(x for x in os.listdir(os.path.join(REF, fam)) if x[0] == "0")] \
    if os.path.isdir(REF) else []


def fixture_name(family, p):
    return f"keras_weights_{family}_{p}"


def build(family, p):
    h = importlib.import_module("deepq-decoding_amd.hdf5_reader")
    d = os.path.join(REF, family, p)
    w = h.read_keras_weights(os.path.join(d, "final_dqn_weights.h5f"))
    out = {f"w{i}": np.ascontiguousarray(x, dtype=np.float32) for i, x in enumerate(w)}
    res = pickle.load(open(os.path.join(d, "all_results.p"), "rb"))
    keys = sorted(res, key=float)
    out["ref_test_p"] = np.array([float(k) for k in keys])
    out["ref_lifetime"] = np.array([float(res[k]) for k in keys])
    cfg = pickle.load(open(glob.glob(os.path.join(d, "variable_config_*.p"))[0], "rb"))
    out["variable_config_keys"] = np.array(sorted(cfg))
    out["variable_config_values"] = np.array([float(cfg[k]) for k in sorted(cfg)])
    return out


# The two agents the reference trained FROM SCRATCH (max_eps = 1, first error rate of their family): their training_history.json is the reference's
# own record of a whole DQNAgent.fit run -- per episode: loss, mean_q, mean_eps (nan-means over the episode's trained steps), episode_reward,
# nb_episode_steps, nb_steps (cumulative), episode_lifetimes_rolling_avg (over rolling_average_length = 1000 episodes) -- together with the
# hyper-parameters that produced it (fixed_config.p, variable_config_N.p) and the lifetimes it tested at (all_results.p).  Committed as arrays
# (float32 where the values allow; nb_steps exact) for tests/test_host_logic.py (step arithmetic) and tools/replay_reference_training.py.
HISTORIES = [("d5_x", "0.001"), ("d5_dp", "0.001")]


def history_fixture_name(family, p):
    return f"training_history_{family}_{p}"


def build_history(family, p):
    import json
    d = os.path.join(REF, family, p)
    h = json.load(open(os.path.join(d, "training_history.json")))
    out = {}
    for k in ("loss", "mean_q", "episode_reward", "episode_lifetimes_rolling_avg", "best_rolling_avg", "duration"):
        out[k] = np.array(h[k], dtype=np.float32)
    out["mean_eps"] = np.array(h["mean_eps"], dtype=np.float64)             # (checked to 1e-12 against the annealing formula)
    for k in ("nb_episode_steps", "nb_steps", "best_episode", "time_since_best", "episode"):
        out[k] = np.array(h[k], dtype=np.int64)
    for k in ("has_succeeded", "stopped_improving"):
        out[k] = np.array(h[k], dtype=np.bool_)
    out["key_order"] = np.array(list(h))
    fixed = pickle.load(open(os.path.join(REF, family, "fixed_config.p"), "rb"))
    var = pickle.load(open(glob.glob(os.path.join(d, "variable_config_*.p"))[0], "rb"))
    out["fixed_config_json"] = np.array(json.dumps(fixed, sort_keys=True))
    out["variable_config_json"] = np.array(json.dumps(var, sort_keys=True))
    out["variable_config_file"] = np.array(os.path.basename(glob.glob(os.path.join(d, "variable_config_*.p"))[0]))
    res = pickle.load(open(os.path.join(d, "all_results.p"), "rb"))
    keys = sorted(res, key=float)
    out["ref_test_p"] = np.array([float(k) for k in keys])
    out["ref_lifetime"] = np.array([float(res[k]) for k in keys])
    return out


def build_history_tails():
    """Of EVERY shipped training_history.json (14 runs): the last five episode rows of the early-stopping bookkeeping and the largest
    time_since_best seen before min_nb_steps -- what pins the fork's stopping rule (tests/test_host_logic.py)."""
    import json
    out = {"agents": np.array([f"{fam}/{p}" for fam, p in AGENTS])}
    rows = []
    for fam, p in AGENTS:
        d = os.path.join(REF, fam, p)
        h = json.load(open(os.path.join(d, "training_history.json")))
        fixed = pickle.load(open(os.path.join(REF, fam, "fixed_config.p"), "rb"))
        var = pickle.load(open(glob.glob(os.path.join(d, "variable_config_*.p"))[0], "rb"))
        S, tsb = np.array(h["nb_steps"]), np.array(h["time_since_best"])
        pre = tsb[S < var["exploration_fraction"]]
        rows.append(dict(patience=fixed["stopping_patience"], min_nb_steps=var["exploration_fraction"], max_timesteps=fixed["max_timesteps"],
                         success_threshold=var["success_threshold"], max_tsb_before_min=int(pre.max()) if len(pre) else 0,
                         max_tsb_after_min_before_last=int(tsb[S >= var["exploration_fraction"]][:-1].max()),
                         tail_nb_steps=[int(x) for x in S[-5:]], tail_time_since_best=[int(x) for x in tsb[-5:]],
                         tail_stopped_improving=[bool(x) for x in h["stopped_improving"][-5:]], tail_has_succeeded=[bool(x) for x in h["has_succeeded"][-5:]],
                         any_stopped_before_last=bool(any(h["stopped_improving"][:-1])), episodes=len(S),
                         max_rolling=float(max(h["episode_lifetimes_rolling_avg"]))))
    out["rows_json"] = np.array(json.dumps(rows, sort_keys=True))
    return out


def main():
    path = os.path.join(ROOT, "tests", "golden", "training_history_tails.npz")
    np.savez_compressed(path, **build_history_tails())
    print(path, os.path.getsize(path))
    for family, p in HISTORIES:
        path = os.path.join(ROOT, "tests", "golden", history_fixture_name(family, p) + ".npz")
        np.savez_compressed(path, **build_history(family, p))
        print(path, os.path.getsize(path))
    for family, p in AGENTS:
        path = os.path.join(ROOT, "tests", "golden", fixture_name(family, p) + ".npz")
        np.savez_compressed(path, **build(family, p))
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
