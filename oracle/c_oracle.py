"""ctypes binding of oracle/env_oracle.c.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libenv_oracle.so")
_lib = None

_u8p = ctypes.POINTER(ctypes.c_uint8)
_u32p = ctypes.POINTER(ctypes.c_uint32)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)


def build(force=False):
    src = os.path.join(_HERE, "env_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.dqo_env_create.restype = ctypes.c_void_p
        _lib.dqo_env_create.argtypes = [ctypes.c_int] * 5 + [ctypes.c_uint32] * 3
        _lib.dqo_env_destroy.argtypes = [ctypes.c_void_p]
        _lib.dqo_env_set_rates.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double]
        _lib.dqo_env_set_referee.argtypes = [ctypes.c_void_p, _u8p, _u8p]
        _lib.dqo_env_num_actions.argtypes = [ctypes.c_void_p]
        _lib.dqo_env_obs_size.argtypes = [ctypes.c_void_p]
        _lib.dqo_env_reset.argtypes = [ctypes.c_void_p, _u8p, _u8p, _u64p, _u32p]
        _lib.dqo_env_step.argtypes = [ctypes.c_void_p, _i32p, ctypes.c_int, _u8p, _f32p, _u8p, _u64p, _u32p, _u8p]
        _lib.dqo_env_export.argtypes = [ctypes.c_void_p, _u64p, _u64p]
        _lib.dqo_env_poke.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_int]
        _lib.dqo_policy_uniform_legal.argtypes = [ctypes.c_void_p, ctypes.c_uint64, _u64p, _i32p]
        _lib.dqo_build_lut.argtypes = [ctypes.c_int, ctypes.c_int, _u8p]
        _lib.dqo_philox.argtypes = [_u32p, _u32p, _u32p]
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def build_lut(d, typ):
    n = (d * d - 1) // 2
    out = np.zeros(1 << n, dtype=np.uint8)
    assert lib().dqo_build_lut(d, typ, _p(out, _u8p)) == 0
    return out


_LUTS = {}


def luts(d):
    if d not in _LUTS:
        _LUTS[d] = (build_lut(d, 3), build_lut(d, 1))
    return _LUTS[d]


class COracleEnv:
    """Batched CPU environment with the same call shape as the product's VectorEnv."""

    def __init__(self, d=5, p_phys=0.01, p_meas=0.01, error_model="DP", use_Y=True, volume_depth=3,
                 n_envs=1, env_id_base=0, seed=(0x5EED, 0xD0DEC0DE), lut=None):
        self.L = lib()
        self.d, self.n_envs, self.depth = d, n_envs, volume_depth
        model = {"X": 0, "DP": 1, "IIDXZ": 2}[error_model]
        self.h = self.L.dqo_env_create(d, model, int(use_Y), volume_depth, n_envs, env_id_base, seed[0], seed[1])
        if not self.h:
            raise ValueError("unsupported configuration")
        self.lut_x, self.lut_z = lut if lut is not None else luts(d)
        self.L.dqo_env_set_referee(self.h, _p(self.lut_x, _u8p), _p(self.lut_z, _u8p))
        self.set_rates(p_phys, p_meas)
        self.num_actions = self.L.dqo_env_num_actions(self.h)
        self.obs_size = self.L.dqo_env_obs_size(self.h)
        n = 2 * d + 1
        self.obs_shape = (self.obs_size // (n * n), n, n)
        self.obs = np.zeros((n_envs,) + self.obs_shape, dtype=np.uint8)
        self.reward = np.zeros(n_envs, dtype=np.float32)
        self.done = np.zeros(n_envs, dtype=np.uint8)
        self.legal = np.zeros((n_envs, 2), dtype=np.uint64)
        self.lifetime = np.zeros(n_envs, dtype=np.uint32)
        self.was_reset = np.zeros(n_envs, dtype=np.uint8)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.dqo_env_destroy(self.h)
            self.h = None

    def set_rates(self, p_phys, p_meas):
        self.L.dqo_env_set_rates(self.h, float(p_phys), float(p_meas))

    def reset(self, which=None):
        w = None if which is None else np.ascontiguousarray(which, dtype=np.uint8)
        self.L.dqo_env_reset(self.h, _p(w, _u8p), _p(self.obs, _u8p), _p(self.legal, _u64p), _p(self.lifetime, _u32p))
        if which is None:
            self.done[:] = 0
        else:
            self.done[w != 0] = 0
        return self.obs

    def step(self, action, auto_reset=False, want_obs=True):
        a = np.ascontiguousarray(action, dtype=np.int32)
        self.L.dqo_env_step(self.h, _p(a, _i32p), int(auto_reset), _p(self.obs if want_obs else None, _u8p),
                            _p(self.reward, _f32p), _p(self.done, _u8p), _p(self.legal, _u64p),
                            _p(self.lifetime, _u32p), _p(self.was_reset, _u8p))
        return self.obs, self.reward, self.done

    def export(self):
        st = np.zeros((self.n_envs, 8), dtype=np.uint64)
        vol = np.zeros((self.n_envs, self.depth), dtype=np.uint64)
        self.L.dqo_env_export(self.h, _p(st, _u64p), _p(vol, _u64p))
        return dict(xmask=st[:, 0], zmask=st[:, 1], true_word=st[:, 2], summed=st[:, 3], acted=st[:, 4],
                    round=st[:, 5], completed=st[:, 6:8], volume=vol)

    def poke(self, i, xmask, zmask, done=0):
        self.L.dqo_env_poke(self.h, i, int(xmask), int(zmask), int(done))

    def policy_uniform_legal(self, t):
        a = np.zeros(self.n_envs, dtype=np.int32)
        self.L.dqo_policy_uniform_legal(self.h, int(t), _p(self.legal, _u64p), _p(a, _i32p))
        return a
