import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@pytest.fixture(scope="session")
def dq():
    """The product package (its directory name has a hyphen, so it is imported by string)."""
    return importlib.import_module("deepq-decoding_amd")


TRACES = ["c1_d3_x", "c2_d5_x", "c3_d5_dp", "c5_d7_dp", "x1_d5_dpy", "x2_d5_dp_hot", "x3_d3_x_nomeas", "x4_d7_x", "x5_d5_iidxz", "x6_d7_iidxz", "x7_d3_dpy"]
STICKY_TRACES = ["sticky_c3_d5_dp", "sticky_x2_d5_dp_hot"]
# lattices beyond one 64-bit word per plane (wide environment + matching referee): legal / acted are stored as word arrays
BIG_TRACES = ["b1_d9_dp", "b2_d9_x", "b3_d11_dp", "b4_d13_x", "b5_d15_dpy", "b6_d9_iidxz"]


def trace_config(g):
    d, model, use_Y, depth, n_envs, n_steps = (int(x) for x in g["config"])
    return dict(d=d, error_model={0: "X", 1: "DP", 2: "IIDXZ"}[model], use_Y=bool(use_Y), volume_depth=depth,
                p_phys=float(g["rates"][0]), p_meas=float(g["rates"][1])), n_envs, n_steps, tuple(int(x) for x in g["seed"])
