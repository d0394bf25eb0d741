"""CPU-only: the C-ABI library builds, loads, and exports every symbol include/deepq_hip.h declares.
No compute call is made here (there is no GPU in the build container)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def built_lib():
    sys.path.insert(0, os.path.join(ROOT, "deepq-decoding_amd"))
    try:
        import build as dq_build
        return dq_build.build(verbose=False)
    finally:
        sys.path.pop(0)


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "deepq_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dq_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(built_lib):
    names = _declared_functions()
    assert "dq_env_step" in names and "dq_env_reset" in names and len(names) >= 15
    out = subprocess.check_output(["nm", "-D", "--defined-only", built_lib], text=True)
    exported = set(re.findall(r" T (dq_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, f"declared in include/deepq_hip.h but not exported: {missing}"


def test_python_binding_covers_header(dq, built_lib):
    from importlib import import_module
    _lib = import_module("deepq-decoding_amd._lib")
    assert sorted(_lib.SIGNATURES) == _declared_functions()
    L = _lib.lib()
    assert L.dq_version() >= 1
    assert isinstance(L.dq_device_count(), int)


def test_fails_loudly_without_gpu(dq):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(dq.DeepQError):
        dq.require_gpu()
    with pytest.raises(dq.DeepQError):
        dq.VectorEnv(d=3, error_model="X", n_envs=2)


def test_header_is_plain_c(built_lib, tmp_path):
    """include/deepq_hip.h must compile as C (no torch / C++ types in the boundary)."""
    src = tmp_path / "t.c"
    src.write_text('#include "deepq_hip.h"\nint main(void){ dq_env_cfg c; (void)c; return dq_version() < 0; }\n')
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "t.o")])


def test_struct_layouts_match_the_library(dq):
    """Every ctypes structure of the binding has the size the library was compiled with (include/deepq_hip.h dq_struct_size): a by-value or by-pointer
    struct of the wrong size is silent corruption -- the class of bug round 4 found in the ncclUniqueId of dist.py."""
    import importlib
    L = importlib.import_module("deepq-decoding_amd._lib")
    lib = L.lib()
    for i, st in enumerate((L.EnvCfg, L.EnvInfo, L.SampleJob, L.QNetCfg, L.QNetJob, L.TdJob, L.EnvStepJob, L.EnvRing)):
        assert lib.dq_struct_size(i) == ctypes.sizeof(st), (i, st.__name__, lib.dq_struct_size(i), ctypes.sizeof(st))
    assert lib.dq_struct_size(99) == -1


def test_library_is_built_from_these_sources(dq, built_lib):
    """build.py compiles the sources' code digest into the library (dq_build_digest, deepq-decoding_amd/_digest.py): after a build it equals the tree's."""
    from importlib import import_module
    L, D = import_module("deepq-decoding_amd._lib"), import_module("deepq-decoding_amd._digest")
    assert L.lib().dq_build_digest().decode() == D.csrc_digest()


@pytest.mark.gpu
def test_the_library_on_this_box_is_the_one_these_sources_build(dq):
    """the -m gpu suite runs a PREBUILT libdeepq_hip.so (git-ignored, carried to the GPU box) and never recompiles.  The digest compiled into
    it must be the digest of the sources that travelled with it -- checked here WITHOUT building, on whatever .so the process has loaded."""
    from importlib import import_module
    L, D = import_module("deepq-decoding_amd._lib"), import_module("deepq-decoding_amd._digest")
    assert L.lib().dq_build_digest().decode() == D.csrc_digest(), "deepq-decoding_amd/lib/libdeepq_hip.so was built from other kernel sources: run deepq-decoding_amd/build.py"
