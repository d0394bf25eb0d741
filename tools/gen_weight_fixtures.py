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
# (family, training error rate): the DP agent the round-1 behavioural test used, the X-noise agent at the same rate (the (6,11,11) /
# 26-action network), and the DP agent trained at the headline rate of BASELINE.json's configs[2]
AGENTS = [("d5_dp", "0.007"), ("d5_x", "0.007"), ("d5_dp", "0.011")]


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


def main():
    for family, p in AGENTS:
        path = os.path.join(ROOT, "tests", "golden", fixture_name(family, p) + ".npz")
        np.savez_compressed(path, **build(family, p))
        print(path, os.path.getsize(path))


if __name__ == "__main__":
    main()
