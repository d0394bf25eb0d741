"""Host-side mirror of the reference environment over the HIP library.

* ``VectorEnv``  -- N independent lattices stepped by one kernel launch; all tensors stay on the GPU.
* ``Surface_Code_Environment_Multi_Decoding_Cycles`` -- drop-in for the reference class of the same
  name (/root/reference/example_notebooks/Environments.py:10-385, "ENV"): one lattice, numpy/int
  views, same attributes and methods, so keras-rl-style loops and the notebooks' helper calls work
  unchanged.

PyTorch is used for device memory and streams only.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from ._lib import EnvCfg, EnvInfo, check, ptr

DEFAULT_SEED = (0x5EED, 0xD0DEC0DE)
_MODELS = {"X": _lib.DQ_MODEL_X, "DP": _lib.DQ_MODEL_DP, "IIDXZ": _lib.DQ_MODEL_IIDXZ}


class _Space:
    """Stand-in for gym.spaces.Box / Discrete (ENV:78-84 only reads shape / n)."""

    def __init__(self, shape=None, n=None, dtype=np.uint8):
        self.shape, self.n, self.dtype, self.low, self.high = shape, n, dtype, 0, 1


def patch_stride_words(d):
    """Words per row of a patch-word observation array (include/deepq_hip.h dq_env_patch_output): d * d words, rows padded to a power of
    two so that they are 16-byte aligned."""
    n = 4
    while n < d * d:
        n *= 2
    return n


def _static_plane(d):
    """padding_syndrome's decoration (ENV:284-298): the cells of a padded plane that do not hold a syndrome bit."""
    n = 2 * d + 1
    x, y = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    v = (((x == 0) | (x == n - 1)) & (y % 2 == 1)) | (((y == 0) | (y == n - 1)) & (x % 2 == 1)) | ((x % 2 == 1) & (y % 2 == 1) & ((x + y) % 4 == 0))
    return v.astype(np.uint8)


_CONVERT_ROWS = 1 << 15      # observations converted at a time (the int64 temporaries below are 8 bytes per cell: a whole 2^20-transition ring at once
                             # allocated ~7 GB -- the configuration the compact ring is meant for then failed to restore a pickled memory)


def _in_chunks(fn, x, trailing):
    """fn over the rows of x (its last `trailing` dimensions form a row), _CONVERT_ROWS rows at a time."""
    lead = tuple(x.shape[:x.dim() - trailing])
    rows = 1
    for v in lead:
        rows *= int(v)
    if rows <= _CONVERT_ROWS:
        return fn(x)
    flat = x.reshape((rows,) + tuple(x.shape[x.dim() - trailing:]))
    first = fn(flat[:_CONVERT_ROWS])
    out = torch.empty((rows,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
    out[:_CONVERT_ROWS] = first
    for i in range(_CONVERT_ROWS, rows, _CONVERT_ROWS):
        out[i:i + _CONVERT_ROWS] = fn(flat[i:i + _CONVERT_ROWS])
    return out.reshape(lead + tuple(out.shape[1:]))


def patch_to_obs(patch, d, depth, layers):
    """Patch words (int32 tensor [..., stride >= d*d]: dq_env_patch_output) -> the padded uint8 observation [..., depth + layers, 2d+1, 2d+1]
    the reference builds (padding_syndrome / padding_actions, ENV:273-314).  The image is a fixed function of the words: syndrome cell (a, b)
    of plane j is bit 4 j + 2 dy + dx of pixel (a - dy, b - dx), action plane l's cell of qubit p is bit 4 depth + l of pixel p."""
    return _in_chunks(lambda x: _patch_to_obs(x, d, depth, layers), patch, 1)


def _patch_to_obs(patch, d, depth, layers):
    n = 2 * d + 1
    w = patch[..., :d * d].to(torch.int64) & 0xFFFFFFFF
    lead = w.shape[:-1]
    w = w.reshape(lead + (d, d))
    out = torch.zeros(lead + (depth + layers, n, n), dtype=torch.uint8, device=patch.device)
    static = torch.from_numpy(_static_plane(d)).to(patch.device)
    for j in range(depth):
        plane = out[..., j, :, :]
        plane += static
        for a_hi in (0, 1):                      # grid rows 0 .. d-1 from the pixel's upper corners, row d from the last pixel row's lower ones
            for b_hi in (0, 1):
                rows = slice(0, d) if not a_hi else slice(d - 1, d)
                cols = slice(0, d) if not b_hi else slice(d - 1, d)
                bit = (w[..., rows, cols] >> (4 * j + 2 * a_hi + b_hi)) & 1
                ys = slice(0, 2 * d, 2) if not a_hi else slice(2 * d, 2 * d + 1)
                xs = slice(0, 2 * d, 2) if not b_hi else slice(2 * d, 2 * d + 1)
                plane[..., ys, xs] = bit.to(torch.uint8)
    for l in range(layers):
        out[..., depth + l, 1:2 * d:2, 1:2 * d:2] = ((w >> (4 * depth + l)) & 1).to(torch.uint8)
    return out


def obs_to_patch(obs, d, depth, layers, stride=None):
    """The inverse of patch_to_obs on observations the environment can produce: uint8 [..., C, 2d+1, 2d+1] -> int32 [..., stride]."""
    stride = patch_stride_words(d) if stride is None else stride
    return _in_chunks(lambda x: _obs_to_patch(x, d, depth, layers, stride), obs, 3)


def _obs_to_patch(obs, d, depth, layers, stride):
    o = obs.to(torch.int64)
    lead = o.shape[:-3]
    w = torch.zeros(lead + (d, d), dtype=torch.int64, device=obs.device)
    for j in range(depth):
        for dy in (0, 1):
            for dx in (0, 1):
                w |= o[..., j, 2 * dy:2 * dy + 2 * d:2, 2 * dx:2 * dx + 2 * d:2] << (4 * j + 2 * dy + dx)
    for l in range(layers):
        w |= o[..., depth + l, 1:2 * d:2, 1:2 * d:2] << (4 * depth + l)
    out = torch.zeros(lead + (stride,), dtype=torch.int64, device=obs.device)
    out[..., :d * d] = w.reshape(lead + (d * d,))
    out = torch.where(out >= (1 << 31), out - (1 << 32), out)       # the words' bit patterns as int32
    return out.to(torch.int32)


class VectorEnv:
    """Batched environment: lattice i has global id ``env_id_base + i`` (its RNG stream)."""

    def __init__(self, d=5, p_phys=0.01, p_meas=0.01, error_model="DP", use_Y=True, volume_depth=3,
                 n_envs=1, seed=DEFAULT_SEED, env_id_base=0, device=None, referee="lut", backend="auto"):
        """backend: "auto" -- d <= 7: one 64-bit word per bit-plane, look-up referee (csrc/env.hip); d >= 9: the wide environment with the
        matching referee (csrc/env_big.hip, include/deepq_hip.h dq_envb_*) --, or "wide" to force the latter at any d (tests)."""
        if d % 2 != 1:
            raise Exception("for the surface code d must be odd!")          # Function_Library.py:28-29
        if error_model not in _MODELS:
            raise ValueError("specified error model not currently supported!")   # ENV:66-67 (reference only prints)
        _lib.require_gpu()
        self.L = _lib.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.d, self.error_model, self.use_Y, self.volume_depth = d, error_model, bool(use_Y), volume_depth
        self.n_envs, self.seed, self.env_id_base = int(n_envs), (int(seed[0]), int(seed[1])), int(env_id_base)
        cfg = EnvCfg(d, _MODELS[error_model], int(bool(use_Y)), volume_depth, self.n_envs, self.env_id_base,
                     (ctypes.c_uint32 * 2)(*self.seed))
        h = ctypes.c_void_p()
        self.wide = backend == "wide" or referee == "matching" or (backend == "auto" and d > 7)
        self._pfx = "dq_envb_" if self.wide else "dq_env_"
        with torch.cuda.device(self.device):
            check(getattr(self.L, self._pfx + "create")(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        info = EnvInfo()
        self.legal_words = 2
        if self.wide:
            lw = ctypes.c_int()
            check(self.L.dq_envb_get_info(self._h, ctypes.byref(info), ctypes.byref(lw)))
            self.legal_words = lw.value
        else:
            check(self.L.dq_env_get_info(self._h, ctypes.byref(info)))
        self.num_actions, self.n_action_layers = info.num_actions, info.n_action_layers
        self.identity_index = info.identity_index
        self.obs_shape = (info.obs_c, info.obs_h, info.obs_w)
        self.state_words, self.n_stab = info.state_words, info.n_stab
        self.observation_space = _Space(shape=self.obs_shape)
        self.action_space = _Space(n=self.num_actions)
        self._p_phys, self._p_meas = float(p_phys), float(p_meas)
        check(getattr(self.L, self._pfx + "set_rates")(self._h, self._p_phys, self._p_meas))
        dev, n = self.device, self.n_envs
        self.obs = torch.zeros((n,) + self.obs_shape, dtype=torch.uint8, device=dev)
        self.reward = torch.zeros(n, dtype=torch.float32, device=dev)
        self.done = torch.zeros(n, dtype=torch.uint8, device=dev)
        self.legal = torch.zeros((n, self.legal_words), dtype=torch.int64, device=dev)     # uint64 bit masks
        self.lifetime = torch.zeros(n, dtype=torch.int32, device=dev)
        self.was_reset = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._lut = None
        self.mlp_referee, self._mlp_w = False, None
        self.inexact = torch.zeros(n, dtype=torch.uint8, device=dev)         # wide backend: referee fallback used in the last step
        if self.wide:
            if referee not in ("lut", "matching", None):
                raise NotImplementedError("the wide environment (d >= 9) decodes with the built-in matching referee")
        elif referee == "lut":
            with torch.cuda.device(self.device):
                check(self.L.dq_env_build_referee(self._h, self._stream()))
        elif referee == "ml" or (isinstance(referee, tuple) and len(referee) == 2 and referee[0] == "ml"):
            # maximum-likelihood table for independent component flips; default rate = one round's marginal flip probability
            q = float(referee[1]) if isinstance(referee, tuple) else (p_phys if error_model == "X" else 2.0 * p_phys / 3.0)
            self.build_ml_referee(q)
        elif hasattr(referee, "flat_weights") and (self.n_stab > 24 or getattr(referee, "on_device", False)):
            self.set_referee_mlp(referee)            # a Dense stack where no table fits (d = 7): evaluated on the device, every step
        elif hasattr(referee, "predict"):
            self.set_referee_predict(referee)
        elif referee is not None:
            self.set_referee(*referee)

    # -- plumbing -----------------------------------------------------------------------------------
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def close(self):
        if getattr(self, "_h", None):
            getattr(self.L, self._pfx + "destroy")(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- rates are plain mutable attributes in the reference (Single_Point_Training_Script.py:200-201) ---
    @property
    def p_phys(self):
        return self._p_phys

    @p_phys.setter
    def p_phys(self, v):
        self._p_phys = float(v)
        check(getattr(self.L, self._pfx + "set_rates")(self._h, self._p_phys, self._p_meas))

    @property
    def p_meas(self):
        return self._p_meas

    @p_meas.setter
    def p_meas(self, v):
        self._p_meas = float(v)
        check(getattr(self.L, self._pfx + "set_rates")(self._h, self._p_phys, self._p_meas))

    def build_ml_referee(self, q_flip):
        """Installs the maximum-likelihood referee for independent X- / Z-component flips with probability q_flip per qubit."""
        with torch.cuda.device(self.device):
            check(self.L.dq_env_build_referee_ml(self._h, float(q_flip), self._stream()))
        self._lut = None

    def set_referee(self, lut_x, lut_z=None):
        """Install caller tables: uint8 0/1 arrays of 2**((d*d-1)//2) entries (bit order: include/deepq_hip.h)."""
        def pack(a):
            bits = np.packbits(np.asarray(a, dtype=np.uint8), bitorder="little")
            bits = np.concatenate([bits, np.zeros((-len(bits)) % 4, np.uint8)])
            return torch.from_numpy(bits.view(np.int32).copy()).to(self.device)
        self._lut = (pack(lut_x), None if lut_z is None else pack(lut_z))
        check(self.L.dq_env_set_referee(self._h, ptr(self._lut[0]), ptr(self._lut[1])))

    def set_referee_predict(self, decoder, chunk=1 << 18):
        """Install an arbitrary referee object with the reference's protocol -- ``decoder.predict(x[n, (d+1)**2], batch_size=..,
        verbose=0) -> scores[n, n_classes]``, of which the environment only uses the argmax (ENV:144,150) -- by tabulating it ONCE over
        all 2**n_stab syndromes (d <= 5: 16.8 M rows at d = 5, fed in chunks) into the kernel's joint look-up table
        (dq_env_set_referee_joint).  Inside the step the referee then costs one table read, whatever the object computes."""
        n = self.n_stab
        if n > 24:
            raise NotImplementedError("a .predict referee is tabulated over all syndromes: d <= 5 (use the built-in referees at d = 7)")
        d = self.d
        cells = np.array([a * (d + 1) + b for a, b in _measurement_order(d)], dtype=np.int64)     # stabilizer s -> cell of the (d+1)^2 vector
        total = 1 << n
        codes = np.zeros(total, dtype=np.uint8)
        shifts = np.arange(n, dtype=np.uint32)
        for lo in range(0, total, chunk):
            idx = np.arange(lo, min(total, lo + chunk), dtype=np.uint32)
            x = np.zeros((len(idx), (d + 1) ** 2), dtype=np.int64)
            x[:, cells] = (idx[:, None] >> shifts[None, :]) & 1
            try:
                scores = decoder.predict(x, batch_size=len(idx), verbose=0)
            except TypeError:
                scores = decoder.predict(x)
            codes[lo:lo + len(idx)] = np.argmax(np.asarray(scores), axis=1).astype(np.uint8)
        if self.error_model == "X" and codes.max(initial=0) > 1:
            raise ValueError("the bit-flip model has two homology classes; the referee predicted class %d" % codes.max())
        pad = (-total) % 16
        c = np.concatenate([codes, np.zeros(pad, np.uint8)]).astype(np.uint32).reshape(-1, 16)
        words = (c << (2 * np.arange(16, dtype=np.uint32))[None, :]).sum(axis=1, dtype=np.uint64).astype(np.uint32)
        self._lut = (torch.from_numpy(words.view(np.int32).copy()).to(self.device), None)
        check(self.L.dq_env_set_referee_joint(self._h, ptr(self._lut[0])))

    def set_referee_mlp(self, referee):
        """Install a Dense-stack referee (referee.FeedForwardReferee: .dims, .flat_weights()) to be EVALUATED on the device before every
        step (dq_env_set_referee_mlp) -- the reference's own kind of static_decoder (ENV:53,144), for any d <= 7; None uninstalls."""
        if self.wide:
            raise NotImplementedError("the wide environment (d >= 9) decodes with the built-in matching referee")
        if referee is None:
            check(self.L.dq_env_set_referee_mlp(self._h, 0, None, None))
            self.mlp_referee, self._mlp_w = False, None
            return
        dims = [int(x) for x in referee.dims]
        self._mlp_w = torch.from_numpy(np.ascontiguousarray(referee.flat_weights(), dtype=np.float32)).to(self.device)
        arr = (ctypes.c_int32 * len(dims))(*dims)
        check(self.L.dq_env_set_referee_mlp(self._h, len(dims) - 1, arr, ptr(self._mlp_w)))
        self.mlp_referee = True          # DQNCore: this environment's step does not ride on the dense backward

    def referee_classes(self, action):
        """Classes the installed Dense-stack referee predicts for the lattices as they stand after `action` (int32 tensor [n_envs]); the
        lattices are not stepped (dq_env_referee_classes)."""
        out = torch.empty(self.n_envs, dtype=torch.uint8, device=self.device)
        check(self.L.dq_env_referee_classes(self._h, ptr(action), ptr(out), self._stream()))
        return out

    def get_referee(self):
        n = 1 << (self.n_stab // 2)
        lx, lz = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        check(self.L.dq_env_get_referee(self._h, lx.ctypes.data, lz.ctypes.data, n))
        return lx, lz

    # -- compact observations (include/deepq_hip.h dq_env_patch_output) --------------------------------
    @property
    def patch_supported(self):
        """d <= 7 (the one-word-per-plane kernel) and 4 * volume_depth + action layers <= 32 data bits per pixel."""
        return not self.wide and 4 * self.volume_depth + self.n_action_layers <= 32

    @property
    def patch_stride(self):
        return patch_stride_words(self.d)

    def arm_patch_output(self, out_patch):
        """The NEXT reset / step / act_step launch of this handle also writes the lattices' patch words into `out_patch`
        (int32 [n_envs, patch_stride], 16-byte aligned); one call arms one launch."""
        assert not self.wide, "patch words are written by the d <= 7 kernels only (a wide handle never consumes the arming)"
        assert out_patch.dtype == torch.int32 and out_patch.is_cuda and out_patch.is_contiguous() and out_patch.shape == (self.n_envs, self.patch_stride)
        check(self.L.dq_env_patch_output(self._h, ptr(out_patch), self.patch_stride))

    def disarm_patch_output(self):
        """Withdraws arm_patch_output (callers that arm and then fail before their launch: the library itself disarms on every launch attempt)."""
        if not self.wide and getattr(self, "_h", None):
            self.L.dq_env_patch_output(self._h, None, self.patch_stride)

    def _launch(self, fn, *args):
        """check(fn(*args)); an exception on the way (argument conversion included) leaves no armed patch output behind."""
        try:
            check(fn(*args))
        except BaseException:
            self.disarm_patch_output()
            raise

    def patch_to_obs(self, patch):
        return patch_to_obs(patch, self.d, self.volume_depth, self.n_action_layers)

    def obs_to_patch(self, obs):
        return obs_to_patch(obs, self.d, self.volume_depth, self.n_action_layers, self.patch_stride)

    # -- gym protocol, batched ----------------------------------------------------------------------
    def reset(self, which=None, out_obs=None, out_patch=None, write_obs=True):
        """ENV:99-115 for every lattice (or those with which[i] != 0).  Returns the uint8 observation tensor.  out_patch: the patch words
        as well (arm_patch_output); write_obs=False: only them."""
        obs = (self.obs if out_obs is None else out_obs) if write_obs else None
        w = None if which is None else torch.as_tensor(which, dtype=torch.uint8, device=self.device).contiguous()
        if out_patch is not None:
            self.arm_patch_output(out_patch)
        self._launch(getattr(self.L, self._pfx + "reset"), self._h, ptr(w), ptr(obs), ptr(self.legal), ptr(self.lifetime), self._stream())
        if which is None:
            self.done.zero_()
        else:
            self.done.masked_fill_(w != 0, 0)
        return obs

    def step(self, action, auto_reset=False, out_obs=None, out_patch=None):
        """ENV:118-204 for every lattice.  `action`: int32 device tensor [n_envs].  Returns (obs, reward, done)."""
        if not (isinstance(action, torch.Tensor) and action.dtype == torch.int32 and action.is_cuda and action.is_contiguous()):
            action = torch.as_tensor(action, dtype=torch.int32, device=self.device).contiguous()
        obs = self.obs if out_obs is None else out_obs
        if out_patch is not None:
            self.arm_patch_output(out_patch)
        if self.wide:
            check(self.L.dq_envb_step(self._h, ptr(action), int(auto_reset), ptr(obs), ptr(self.reward), ptr(self.done),
                                      ptr(self.legal), ptr(self.lifetime), ptr(self.was_reset), ptr(self.inexact), self._stream()))
        else:
            self._launch(self.L.dq_env_step, self._h, ptr(action), int(auto_reset), ptr(obs), ptr(self.reward), ptr(self.done),
                         ptr(self.legal), ptr(self.lifetime), ptr(self.was_reset), self._stream())
        return obs, self.reward, self.done

    def act_step(self, t, q=None, eps=1.0, masked_greedy=False, auto_reset=True, out_obs=None, out_action=None, out_patch=None):
        """Action selection (the rule of select_actions) fused in front of the step: one launch.  Returns the actions taken."""
        if out_action is None:
            out_action = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        obs = self.obs if out_obs is None else out_obs
        if out_patch is not None:
            self.arm_patch_output(out_patch)
        seed = (ctypes.c_uint32 * 2)(*self.seed)
        if self.wide:
            check(self.L.dq_envb_act_step(self._h, ptr(q), float(eps), int(masked_greedy), seed, int(t), ptr(out_action), int(auto_reset),
                                          ptr(obs), ptr(self.reward), ptr(self.done), ptr(self.legal), ptr(self.lifetime),
                                          ptr(self.was_reset), ptr(self.inexact), self._stream()))
        else:
            self._launch(self.L.dq_env_act_step, self._h, ptr(q), float(eps), int(masked_greedy), seed, int(t), ptr(out_action), int(auto_reset),
                         ptr(obs), ptr(self.reward), ptr(self.done), ptr(self.legal), ptr(self.lifetime), ptr(self.was_reset), self._stream())
        return out_action

    def act_steps(self, n_steps, t0, action_ring, reward_ring=None, done_ring=None, obs_ring=None, patch_ring=None, slot0=0, auto_reset=True):
        """n_steps agent steps of an acting loop under the uniform-over-legal policy in ONE launch (include/deepq_hip.h dq_env_act_steps): step s = policy counter
        t0 + s; its transition goes to slot (slot0 + s) mod T of the rings `action_ring` int32 [T, n], `reward_ring` float32 [T, n], `done_ring` uint8 [T, n], the
        successor observation to the slot behind it of `obs_ring` uint8 [T, n, C, H, W] / `patch_ring` int32 [T, n, stride].  self.legal / lifetime / was_reset hold
        the last step's values; self.obs / reward / done are NOT updated (they live in the rings).  Same bits as n_steps calls of act_step(t0 + s, q=None)."""
        if self.wide:
            raise NotImplementedError("act_steps: not offered by the wide environment")
        T = int(action_ring.shape[0])
        for r in (action_ring, reward_ring, done_ring, obs_ring, patch_ring):
            assert r is None or (r.is_contiguous() and int(r.shape[0]) == T and int(r.shape[1]) == self.n_envs)
        ring = _lib.EnvRing(action_ring_dev=ptr(action_ring), reward_ring_dev=ptr(reward_ring), done_ring_dev=ptr(done_ring), obs_ring_dev=ptr(obs_ring),
                            patch_ring_dev=ptr(patch_ring), patch_stride_words=int(patch_ring.shape[2]) if patch_ring is not None else 0, n_slots=T, slot0=int(slot0) % T)
        seed = (ctypes.c_uint32 * 2)(*self.seed)
        self._launch(self.L.dq_env_act_steps, self._h, int(n_steps), seed, int(t0), ctypes.byref(ring), int(auto_reset), ptr(self.legal), ptr(self.lifetime),
                     ptr(self.was_reset), self._stream())

    def select_actions(self, t, q=None, eps=1.0, masked_greedy=False, out=None):
        """Epsilon-greedy over the legal set on the device (include/deepq_hip.h: dq_policy_select)."""
        if out is None:
            out = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        seed = (ctypes.c_uint32 * 2)(*self.seed)
        if self.wide:
            check(self.L.dq_policy_select_wide(ptr(q), ptr(self.legal), self.n_envs, self.num_actions, self.legal_words, float(eps),
                                               int(masked_greedy), seed, self.env_id_base, int(t), ptr(out), self._stream()))
        else:
            check(self.L.dq_policy_select(ptr(q), ptr(self.legal), self.n_envs, self.num_actions, float(eps), int(masked_greedy),
                                          seed, self.env_id_base, int(t), ptr(out), self._stream()))
        return out

    # -- state views --------------------------------------------------------------------------------------
    def export_state(self):
        """int64 tensor [n_envs, 11 + depth] of uint64 words (layout: include/deepq_hip.h)."""
        st = torch.zeros((self.n_envs, self.state_words), dtype=torch.int64, device=self.device)
        check(getattr(self.L, self._pfx + "export_state")(self._h, ptr(st), self._stream()))
        return st

    def import_state(self, st):
        if self.wide:
            raise NotImplementedError("import_state: not offered by the wide environment")
        st = torch.as_tensor(st, dtype=torch.int64, device=self.device).contiguous()
        assert st.shape == (self.n_envs, self.state_words)
        check(self.L.dq_env_import_state(self._h, ptr(st), self._stream()))

    def tables(self):
        sq, qs, nq = (np.zeros(64, np.uint64) for _ in range(3))
        ty = np.zeros(64, np.uint8)
        check(self.L.dq_env_get_tables(self._h, sq.ctypes.data, qs.ctypes.data, nq.ctypes.data, ty.ctypes.data))
        return dict(stab_qmask=sq, qubit_smask=qs, neigh_qmask=nq, stab_type=ty)


# ---------------------------------------------------------------------------------------------------------
# lattice helpers for the single-lattice facade (host-side table code; init-time / notebook helpers only)
# ---------------------------------------------------------------------------------------------------------

def _plaquette_type(d, a, b):
    if (a == 0 and b % 2 == 0) or (a == d and b % 2 == 1) or (b == 0 and a % 2 == 1) or (b == d and a % 2 == 0):
        return 0
    return 3 if (a + b) % 2 else 1


def generateSurfaceCodeLattice(d):
    """Function_Library.py:13-51."""
    if np.mod(d, 2) != 1:
        raise Exception("for the surface code d must be odd!")
    q = np.zeros((d, d, 4, 3), dtype=np.int64)
    for x in range(d):
        for y in range(d):
            for k, (a, b) in enumerate(((x, y), (x, y + 1), (x + 1, y), (x + 1, y + 1))):
                q[x, y, k] = (a, b, _plaquette_type(d, a, b))
    return q


def _measurement_order(d):
    half = (d + 1) // 2 - 1
    return ([(a, b) for a in range(1, d) for b in range(1, d)] + [(0, 2 * x + 1) for x in range(half)] +
            [(d, 2 * x + 2) for x in range(half)] + [(2 * x + 2, 0) for x in range(half)] + [(2 * x + 1, d) for x in range(half)])


def _u64(x):
    return int(x) & 0xFFFFFFFFFFFFFFFF


class Surface_Code_Environment_Multi_Decoding_Cycles:
    """Drop-in for ENV:10-385 running on the GPU (one lattice).

    Differences from the reference, all forced by what the checkout lacks:
      * randomness comes from the site-indexed Philox stream (``seed``, ``env_id``) instead of numpy's
        unseeded global generator;
      * ``static_decoder`` is the built-in minimum-weight look-up referee (None / "lut"), the maximum-likelihood one
        ("ml" or ("ml", q_flip)), a pair of 0/1 tables, or -- as in the reference -- any object with ``.predict`` (d <= 5): it is
        tabulated once over all syndromes (VectorEnv.set_referee_predict), since only the argmax of its output is used (ENV:150).
    """

    def __init__(self, d=5, p_phys=0.01, p_meas=0.01, error_model="DP", use_Y=True, volume_depth=3, static_decoder=None,
                 seed=DEFAULT_SEED, env_id=0, device=None):
        if static_decoder is None or static_decoder == "lut" or static_decoder is True:
            referee = "lut"
        elif static_decoder == "ml":
            referee = "ml"
        elif isinstance(static_decoder, (tuple, list)):
            referee = tuple(static_decoder)
        elif hasattr(static_decoder, "predict"):
            referee = static_decoder            # the reference's protocol (ENV:144): tabulated once over all syndromes (d <= 5)
        elif static_decoder == "matching":
            referee = "matching"
        else:
            raise NotImplementedError("static_decoder must be None/'lut', 'ml', a (lut_x, lut_z) pair or an object with .predict")
        self._v = VectorEnv(d, p_phys, p_meas, error_model, use_Y, volume_depth, n_envs=1, seed=seed, env_id_base=env_id,
                            device=device, referee=referee)
        v = self._v
        self.d, self.error_model, self.use_Y, self.volume_depth, self.static_decoder = d, error_model, use_Y, volume_depth, static_decoder
        self.num_actions, self.n_action_layers, self.identity_index = v.num_actions, v.n_action_layers, v.identity_index
        self.identity_indicator = self.generate_identity_indicator(d)
        self.qubits = generateSurfaceCodeLattice(d)
        self.qubit_stabilizers = self.get_stabilizer_list(self.qubits, d)
        self.qubit_neighbours = self.get_qubit_neighbour_list(d)
        self.observation_space, self.action_space = v.observation_space, v.action_space
        self.board_state = np.zeros(v.obs_shape, dtype=np.int64)
        self.done = False
        self.lifetime = 0
        self.multi_cycle = True                                              # ENV:97
        self._order = _measurement_order(d)
        self._action = torch.zeros(1, dtype=torch.int32, device=v.device)
        self._state = None

    # rates ------------------------------------------------------------------------------------------------
    p_phys = property(lambda self: self._v.p_phys, lambda self, x: setattr(self._v, "p_phys", x))
    p_meas = property(lambda self: self._v.p_meas, lambda self, x: setattr(self._v, "p_meas", x))

    # gym protocol --------------------------------------------------------------------------------------------
    def _pull(self):
        self.board_state[...] = self._v.obs[0].cpu().numpy()               # same ndarray object every call (ENV:115,204)
        self.done = bool(self._v.done[0].item())
        self.lifetime = int(self._v.lifetime[0].item())
        self._state = None

    def reset(self):
        self._v.reset()
        self._pull()
        return self.board_state

    def step(self, action):
        action = int(action)
        if not 0 <= action < self.num_actions:
            raise IndexError(f"index {action} is out of bounds for axis 0 with size {self.num_actions}")   # ENV:131
        self._action.fill_(action)
        self._v.step(self._action)
        reward = float(self._v.reward[0].item())
        self._pull()
        return self.board_state, reward, self.done, {}

    def initialize_state(self):
        """ENV:206-235 (also resets legal moves on the device; the reference leaves them stale until reset())."""
        self._v.reset()
        self._pull()

    def reset_legal_moves(self):
        """ENV:238-258: forget the moves made in this volume; legal = identity + every action on a qubit that touches a stabilizer
        which fired anywhere in the current (faulty) volume.  The bookkeeping lives on the device: the lattice record is exported,
        rewritten and imported back (a helper call, not part of the stepped path -- reset() / step() do this inside the kernel)."""
        v, d2 = self._v, self.d * self.d
        if v.wide:
            raise NotImplementedError("reset_legal_moves: not offered on lattices beyond d = 7 (reset() / step() do it inside the kernel)")
        st = v.export_state()
        w = [_u64(x) for x in st[0].cpu().tolist()]
        fired = 0
        for x in w[11:11 + self.volume_depth]:
            fired |= x
        qubit_smask = v.tables()["qubit_smask"]
        legal = 1 << self.identity_index
        for q in range(d2):
            if int(qubit_smask[q]) & fired:
                for j in range(self.n_action_layers):
                    legal |= 1 << (q + j * d2)
        w[4] = w[6] = w[7] = 0
        w[8], w[9] = legal & 0xFFFFFFFFFFFFFFFF, legal >> 64
        signed = [x - (1 << 64) if x >= (1 << 63) else x for x in w]
        v.import_state(torch.tensor([signed], dtype=torch.int64))
        v.legal.copy_(torch.tensor([[signed[8], signed[9]]], dtype=torch.int64))
        self._state = None

    # state views in the reference's data types ----------------------------------------------------------------
    def _words(self):
        if self._state is None:
            self._state = [_u64(x) for x in self._v.export_state()[0].cpu().tolist()]
        return self._state

    def _fields(self):
        """The exported record as Python integers (bit q / s / a = qubit / stabilizer / action), whatever the backend's word layout
        (include/deepq_hip.h: dq_env_export_state, dq_envb_export_state)."""
        w = self._words()

        def big(words):
            return sum(x << (64 * k) for k, x in enumerate(words))

        if self._v.wide:
            W, LW = (self.d * self.d + 63) // 64, self._v.legal_words
            o = 5 * W + 1
            return dict(x=big(w[0:W]), z=big(w[W:2 * W]), true=big(w[2 * W:3 * W]), acted=big(w[4 * W:5 * W]),
                        comp=big(w[o:o + LW]), legal=big(w[o + LW:o + 2 * LW]),
                        vol=[big(w[o + 2 * LW + 1 + j * W:o + 2 * LW + 1 + (j + 1) * W]) for j in range(self.volume_depth)])
        return dict(x=w[0], z=w[1], true=w[2], acted=w[4], comp=big(w[6:8]), legal=big(w[8:10]), vol=w[11:11 + self.volume_depth])

    @staticmethod
    def _mask_to_set(mask):
        return {a for a in range(mask.bit_length()) if (mask >> a) & 1}

    @property
    def legal_actions(self):
        return self._mask_to_set(self._fields()["legal"])

    @property
    def acted_on_qubits(self):
        return self._mask_to_set(self._fields()["acted"])

    @property
    def completed_actions(self):
        done = self._fields()["comp"]
        return np.array([(done >> a) & 1 for a in range(self.num_actions)], dtype=int)

    @property
    def hidden_state(self):
        f, d = self._fields(), self.d
        out = np.zeros(d * d)
        for q in range(d * d):
            out[q] = (1 if (f["x"] >> q) & 1 else 0) ^ (3 if (f["z"] >> q) & 1 else 0)
        return out.reshape(d, d)

    def _word_to_grid(self, word):
        g = np.zeros((self.d + 1, self.d + 1), dtype=int)
        for s, (a, b) in enumerate(self._order):
            g[a, b] = (word >> s) & 1
        return g

    @property
    def current_true_syndrome(self):
        return self._word_to_grid(self._fields()["true"])

    @property
    def summed_syndrome_volume(self):
        return sum(self._word_to_grid(x) for x in self._fields()["vol"])

    def is_adjacent_to_syndrome(self, qubit_number):
        s = self.summed_syndrome_volume
        return any(s[st] != 0 for st in self.qubit_stabilizers[qubit_number])   # ENV:262-271

    # embedding helpers used by the notebooks' "production decoding" demo (ENV:273-324) -------------------------
    def padding_syndrome(self, syndrome_in):
        n = 2 * self.d + 1
        out = np.zeros((n, n), int)
        out[0, 1::2] = out[n - 1, 1::2] = 1
        out[1::2, 0] = out[1::2, n - 1] = 1
        for x in range(1, n, 2):
            for y in range(1, n, 2):
                if (x + y) % 4 == 0:
                    out[x, y] = 1
        out[0::2, 0::2] = np.asarray(syndrome_in)
        return out

    def padding_actions(self, actions_in):
        n = 2 * self.d + 1
        out = np.zeros((n, n), int)
        for i, taken in enumerate(actions_in):
            if taken:
                out[2 * (i // self.d) + 1, 2 * (i % self.d) + 1] = 1
        return out

    def indicate_identity(self, board_state):
        for k in range(self.n_action_layers):
            board_state[self.volume_depth + k] = board_state[self.volume_depth + k] + self.identity_indicator
        return board_state

    def get_qubit_stabilizer_list(self, qubits, qubit):
        row, col = qubit
        return [tuple(qubits[row, col, j, :2]) for j in range(4) if qubits[row, col, j, 2] != 0]

    def get_stabilizer_list(self, qubits, d):
        return [self.get_qubit_stabilizer_list(qubits, [r, c]) for r in range(d) for c in range(d)]

    def get_qubit_neighbour_list(self, d):
        out = []
        for row in range(d):
            for col in range(d):
                cells = [(row + a, col + b) for a in (0, -1, 1) for b in (0, -1, 1)][1:]
                out.append([r * d + c for (r, c) in cells if 0 <= r < d and 0 <= c < d])
        return out

    def generate_identity_indicator(self, d):
        ind = np.ones((2 * d + 1, 2 * d + 1), int)
        ind[1::2, 1::2] = 0
        return ind
