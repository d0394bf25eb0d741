"""Host wrapper of the HIP Q-network and DQN-update kernels (include/deepq_hip.h: dq_qnet_*, dq_td_*,
dq_adam_step, dq_replay_sample).  Mirrors what the reference gets from Keras + keras-rl:
build_convolutional_nn (/root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:61-90),
the dueling head and the double-DQN train_on_batch of DQNAgent (:119-130)."""
import ctypes

import numpy as np
import torch

from . import _lib, _philox
from ._lib import QNetCfg, QNetJob, check, ptr


TD_METRICS_FLOATS = 2050      # include/deepq_hip.h DQ_TD_METRICS_FLOATS


def _seed_arr(seed):
    return (ctypes.c_uint32 * 2)(int(seed[0]) & 0xFFFFFFFF, int(seed[1]) & 0xFFFFFFFF)


class QNetwork:
    """Conv dueling Q-network with caller-owned flat parameter tensors (Keras order / shapes)."""

    def __init__(self, input_shape, c_layers, ff_layers, n_actions, dueling=True, max_batch=4096, device=None):
        _lib.require_gpu()
        self.L = _lib.lib()
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.input_shape = tuple(int(x) for x in input_shape)
        self.c_layers = [[int(x) for x in l] for l in c_layers]
        self.ff_layers = [[int(l[0]), float(l[1])] for l in ff_layers]
        self.n_actions, self.dueling, self.max_batch = int(n_actions), bool(dueling), int(max_batch)
        cfg = QNetCfg()
        cfg.in_c, cfg.in_h, cfg.in_w = self.input_shape
        cfg.n_conv = len(self.c_layers)
        for i, l in enumerate(self.c_layers):
            for j in range(3):
                cfg.conv[i][j] = l[j]
        cfg.n_ff = len(self.ff_layers)
        for i, l in enumerate(self.ff_layers):
            cfg.ff_units[i], cfg.ff_dropout[i] = l[0], l[1]
        cfg.n_actions, cfg.dueling, cfg.max_batch = self.n_actions, int(self.dueling), self.max_batch
        h = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(self.L.dq_qnet_create(ctypes.byref(cfg), ctypes.byref(h)))
        self._h = h
        self.fused_enabled = True
        self.n_params = int(self.L.dq_qnet_param_count(self._h))
        self.n_conv_params = int(self.L.dq_qnet_conv_param_count(self._h))     # flat layout: convolutions first, then the dense layers
        self.packed_bytes = int(self.L.dq_qnet_packed_bytes(self._h))          # 0 when the fused chains do not cover the architecture
        self.layers = []
        for i in range(self.L.dq_qnet_num_layers(self._h)):
            ko, bo, shape, nd = ctypes.c_int64(), ctypes.c_int64(), (ctypes.c_int32 * 4)(), ctypes.c_int32()
            check(self.L.dq_qnet_layer_info(self._h, i, ctypes.byref(ko), ctypes.byref(bo), ctypes.byref(shape), ctypes.byref(nd)))
            kshape = tuple(shape[j] for j in range(nd.value))
            self.layers.append(dict(kernel_offset=ko.value, bias_offset=bo.value, kernel_shape=kshape, bias_shape=(kshape[-1],)))

    def set_fused(self, enable):
        """Select the fused LDS-resident chains (default) or the per-layer implicit-GEMM path."""
        check(self.L.dq_qnet_set_fused(self._h, 1 if enable else 0))
        self.fused_enabled = bool(enable)

    def set_grad_scale(self, grad_scale):
        """Declare the loss scale a caller-supplied dq was computed with (td_update's grad_scale); 0 = undeclared (the backward then
        measures max |dq| on the device).  See include/deepq_hip.h dq_qnet_set_grad_scale."""
        check(self.L.dq_qnet_set_grad_scale(self._h, float(grad_scale)))

    def mark_conv_backward(self, event):
        """The next fused backward records `event` (torch.cuda.Event) right behind its convolutional kernel's launch
        (include/deepq_hip.h dq_qnet_mark_conv_backward); None clears a pending mark."""
        if event is not None and not event.cuda_event:
            event.record(torch.cuda.current_stream(self.device))    # (torch creates the hipEvent_t lazily, at the first record)
        check(self.L.dq_qnet_mark_conv_backward(self._h, ctypes.c_void_p(event.cuda_event) if event is not None else None))

    def set_patch_input(self, n_syndrome_planes, stride_words):
        """Declare the observation's plane structure so that jobs may read patch words (include/deepq_hip.h dq_qnet_set_patch_input,
        dq_env_patch_output; `patch=True` in a forward_multi job) instead of padded uint8 images.  Call before pack()."""
        check(self.L.dq_qnet_set_patch_input(self._h, int(n_syndrome_planes), int(stride_words)))
        self.patch_planes = int(n_syndrome_planes)

    def set_kernel_forms(self, conv_forward=None, conv_backward=None, conv_backward_a1=None):
        """Which form of the convolution kernels this handle runs (include/deepq_hip.h dq_qnet_set_kernel_forms): conv_forward "wave" | "group", conv_backward
        "default" | "8" | "16", conv_backward_a1 "recompute" | "saved"; None keeps the current choice (initially DQ_CONV_FORM / DQ_CONV_BWD_FORM / DQ_CONV_BWD_A1
        as they were when the handle was created)."""
        f = {None: -1, "wave": 0, "group": 1}[conv_forward]
        b = {None: -1, "default": 0, "8": 1, "16": 2}[conv_backward]
        a1 = {None: -1, "recompute": 0, "saved": 1}[conv_backward_a1]
        check(self.L.dq_qnet_set_kernel_forms(self._h, f, b, a1))

    def check_range(self):
        """Synchronises the current stream; raises DeepQError(status=DQ_ERR_RANGE) if a gradient of the fused backward left the range of
        its f16 pieces since the last call (include/deepq_hip.h dq_qnet_range_check)."""
        check(self.L.dq_qnet_range_check(self._h, self._stream()))

    def range_discarded(self):
        """Optimizer steps the range guard discarded whole since the last call (include/deepq_hip.h dq_qnet_range_discarded); synchronises."""
        n = ctypes.c_uint(0)
        check(self.L.dq_qnet_range_discarded(self._h, ctypes.byref(n), self._stream()))
        return int(n.value)

    @property
    def fused_supported(self):
        return bool(self.L.dq_qnet_fused_supported(self._h))

    def close(self):
        if getattr(self, "_h", None):
            self.L.dq_qnet_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    # -- parameters ---------------------------------------------------------------------------------------
    def init_params(self, seed, stream_id=_lib.STREAM_INIT):
        """Keras defaults: glorot_uniform kernels, zero biases.  Element i of layer l uses
        u = (word + 0.5)/2^32, word = Philox(key=seed, ctr=(i>>2, 0, l, INIT<<16))[i&3]."""
        flat = np.zeros(self.n_params, dtype=np.float32)
        for l, info in enumerate(self.layers):
            k = info["kernel_shape"]
            nk = int(np.prod(k))
            fan_in, fan_out = (k[0] * k[1] * k[2], k[0] * k[1] * k[3]) if len(k) == 4 else k
            limit = np.sqrt(6.0 / (fan_in + fan_out))
            idx = np.arange(nk, dtype=np.uint64)
            words = _philox.philox4x32((idx >> np.uint64(2)).astype(np.uint32), 0, l, stream_id << 16, seed)
            w = np.stack(words, axis=1)[np.arange(nk), (idx & np.uint64(3)).astype(np.int64)]
            u = (w.astype(np.float64) + 0.5) / 4294967296.0
            flat[info["kernel_offset"]:info["kernel_offset"] + nk] = ((2.0 * u - 1.0) * limit).astype(np.float32)
        return torch.from_numpy(flat).to(self.device)

    def get_weights(self, params):
        """List of numpy arrays [kernel, bias, ...] in Keras order (what model.get_weights() returns)."""
        p = params.detach().cpu().numpy()
        out = []
        for info in self.layers:
            nk = int(np.prod(info["kernel_shape"]))
            out.append(p[info["kernel_offset"]:info["kernel_offset"] + nk].reshape(info["kernel_shape"]).copy())
            out.append(p[info["bias_offset"]:info["bias_offset"] + info["bias_shape"][0]].copy())
        return out

    def set_weights(self, params, weights):
        flat = params.detach().cpu().numpy().copy()
        assert len(weights) == 2 * len(self.layers)
        for i, info in enumerate(self.layers):
            k, b = np.asarray(weights[2 * i], np.float32), np.asarray(weights[2 * i + 1], np.float32)
            assert k.shape == info["kernel_shape"] and b.shape == info["bias_shape"], (k.shape, info["kernel_shape"])
            flat[info["kernel_offset"]:info["kernel_offset"] + k.size] = k.reshape(-1)
            flat[info["bias_offset"]:info["bias_offset"] + b.size] = b
        params.copy_(torch.from_numpy(flat))

    def pack(self, params, out=None):
        """f16 pieces of the weights in matrix-core operand order (dq_qnet_pack): pass the result as `packed=` to forward() /
        forward_multi() jobs that use `params`, and call it again whenever `params` changes.  None if not applicable."""
        if not self.packed_bytes:
            return None
        if out is None:
            out = torch.empty(self.packed_bytes, dtype=torch.uint8, device=self.device)
        check(self.L.dq_qnet_pack(self._h, ptr(params), ptr(out), self._stream()))
        return out

    # -- compute ----------------------------------------------------------------------------------------------
    def forward(self, params, obs, batch=None, index=None, index_off=0, index_mod=0, training=False, seed=(0, 0), t=0,
                sample_base=0, out=None, packed=None):
        if packed is not None:
            return self.forward_multi([dict(params=params, obs=obs, batch=batch, index=index, index_off=index_off, index_mod=index_mod,
                                            training=training, seed=seed, t=t, sample_base=sample_base, out=out, packed=packed)])[0]
        assert params.dtype == torch.float32 and params.is_cuda and params.numel() == self.n_params
        assert obs.dtype == torch.uint8 and obs.is_cuda and obs.is_contiguous()
        if batch is None:
            batch = obs.shape[0] if index is None else index.shape[0]
        if index is not None:
            assert index.dtype == torch.int32 and index.is_cuda and index.is_contiguous()
        if out is None:
            out = torch.empty((batch, self.n_actions), dtype=torch.float32, device=self.device)
        check(self.L.dq_qnet_forward(self._h, ptr(params), ptr(obs), ptr(index), int(index_off), int(index_mod), int(batch),
                                     int(bool(training)), _seed_arr(seed), int(t), int(sample_base), ptr(out), self._stream()))
        if training:
            self._train_inputs = (obs, index)       # backward re-reads them (conv1 weight gradient): keep them alive
        return out

    def forward_multi(self, jobs):
        """Several forwards in one pair of launches (dq_qnet_forward_multi).  `jobs`: list of dicts with the keyword arguments
        of forward() (params, obs, batch, index, index_off, index_mod, training, seed, t, sample_base, out) and optionally
        patch=True (obs = int32 patch words, set_patch_input(); all jobs of a call in the same form); at most one training job.
        Returns the list of outputs."""
        arr = (QNetJob * len(jobs))()
        outs = []
        for jb, kw in zip(arr, jobs):
            params, obs, index = kw["params"], kw["obs"], kw.get("index")
            assert params.dtype == torch.float32 and params.is_cuda and params.numel() == self.n_params
            patch = bool(kw.get("patch", False))       # obs holds patch words (int32 [rows, stride]) instead of uint8 images
            assert obs.dtype == (torch.int32 if patch else torch.uint8) and obs.is_cuda and obs.is_contiguous()
            batch = kw.get("batch")
            if batch is None:
                batch = obs.shape[0] if index is None else index.shape[0]
            if index is not None:
                assert index.dtype == torch.int32 and index.is_cuda and index.is_contiguous()
            jb.reserved = 1 if patch else 0
            out = kw.get("out")
            if out is None:
                out = torch.empty((batch, self.n_actions), dtype=torch.float32, device=self.device)
            seed = kw.get("seed", (0, 0))
            jb.params_dev, jb.obs_dev, jb.index_dev = ptr(params), ptr(obs), ptr(index)
            jb.index_off, jb.index_mod, jb.batch = int(kw.get("index_off", 0)), int(kw.get("index_mod", 0)), int(batch)
            jb.training = int(bool(kw.get("training", False)))
            jb.seed[0], jb.seed[1] = int(seed[0]) & 0xFFFFFFFF, int(seed[1]) & 0xFFFFFFFF
            jb.t, jb.sample_base, jb.q_dev = int(kw.get("t", 0)), int(kw.get("sample_base", 0)), ptr(out)
            jb.packed_dev = ptr(kw.get("packed"))
            if jb.training:
                self._train_inputs = (obs, index)
            outs.append(out)
        check(self.L.dq_qnet_forward_multi(self._h, len(jobs), arr, self._stream()))
        return outs

    def backward(self, params, dq, grads=None):
        assert dq.dtype == torch.float32 and dq.is_cuda and dq.is_contiguous()
        if grads is None:
            grads = torch.empty(self.n_params, dtype=torch.float32, device=self.device)
        check(self.L.dq_qnet_backward(self._h, ptr(params), ptr(dq), ptr(grads), self._stream()))
        return grads


def _backward_phase(self, params, dq, grads, phase):
    """Phase 0: dueling + dense layers (grads[n_conv_params:] complete); phase 1: convolutions (grads[:n_conv_params])."""
    check(self.L.dq_qnet_backward_phase(self._h, ptr(params), ptr(dq), ptr(grads), int(phase), self._stream()))
    return grads


QNetwork.backward_phase = _backward_phase


def _backward_adam(self, params, dq, grads, m, v, t, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    """backward() + adam_step() with the optimizer step applied by the backward's last launch (single-process training)."""
    check(self.L.dq_qnet_backward_adam(self._h, ptr(params), ptr(dq), ptr(grads), ptr(m), ptr(v), float(lr), float(beta_1), float(beta_2),
                                       float(epsilon), int(t), self._stream()))
    return grads


QNetwork.backward_adam = _backward_adam


def _td_job(td):
    B, A = td["q_s0"].shape
    j = _lib.TdJob()
    j.q_online_s1_dev, j.q_target_s1_dev, j.q_s0_dev = ptr(td["q_online_s1"]), ptr(td["q_target_s1"]), ptr(td["q_s0"])
    j.reward_dev, j.terminal_dev, j.action_dev, j.index_dev = ptr(td["reward"]), ptr(td["terminal"]), ptr(td["action"]), ptr(td.get("index"))
    j.gamma, j.grad_scale, j.batch, j.n_actions = float(td["gamma"]), float(td.get("grad_scale") or 1.0 / B), B, A
    j.y_dev, j.dq_dev, j.metrics_dev = ptr(td.get("y")), ptr(td.get("dq")), ptr(td.get("metrics"))
    j.auto_scale = int(bool(td.get("auto_scale", False)))
    st = td.get("step_stats")
    if st is not None:
        done, was_reset, lifetime, step_reward, n, stats = st
        j.done_dev, j.was_reset_dev, j.lifetime_dev, j.step_reward_dev, j.n, j.stats_dev = (ptr(done), ptr(was_reset), ptr(lifetime),
                                                                                            ptr(step_reward), int(n), ptr(stats))
    return j


def _td_backward_adam(self, params, td, grads, m, v, t, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    """td_update() (+ episode bookkeeping) + backward() + adam_step() with the TD step computed by the backward's first kernel and the
    optimizer step applied by its last one.  td: dict with q_online_s1, q_target_s1, q_s0, reward, terminal, action, index, gamma,
    grad_scale and optional y, dq, metrics, step_stats = (done, was_reset, lifetime, reward, n, stats)."""
    j = _td_job(td)
    check(self.L.dq_qnet_td_backward_adam(self._h, ptr(params), ctypes.byref(j), ptr(grads), ptr(m), ptr(v), float(lr), float(beta_1),
                                          float(beta_2), float(epsilon), int(t), self._stream()))
    return grads


def _env_step_job(step):
    """dq_env_step_job from a dict with q, eps, masked_greedy, seed, t, action, auto_reset, obs, reward, done, legal, lifetime, was_reset
    and optional sample (a _lib.SampleJob), stats."""
    e = _lib.EnvStepJob()
    e.q_dev, e.eps, e.masked_greedy = ptr(step["q"]), float(step["eps"]), int(step["masked_greedy"])
    e.seed[0], e.seed[1] = int(step["seed"][0]) & 0xFFFFFFFF, int(step["seed"][1]) & 0xFFFFFFFF
    e.t, e.action_dev, e.auto_reset = int(step["t"]), ptr(step["action"]), int(step.get("auto_reset", 1))
    e.obs_dev, e.reward_dev, e.done_dev = ptr(step["obs"]), ptr(step["reward"]), ptr(step["done"])
    e.legal_dev, e.lifetime_dev, e.was_reset_dev = ptr(step["legal"]), ptr(step["lifetime"]), ptr(step["was_reset"])
    sj = step.get("sample")
    e.sample = ctypes.pointer(sj) if sj is not None else None
    e.stats_dev = ptr(step.get("stats"))
    return e


def _td_backward_adam_env(self, params, td, grads, m, v, t, lr, beta_1, beta_2, epsilon, env_handle, step):
    """td_backward_adam() with the SAME vector step's environment launch (dq_env_act_step(_sample) + its episode bookkeeping) riding on
    the dense backward's first kernel (dq_qnet_td_backward_adam_env).  step: see _env_step_job."""
    j = _td_job(td)
    assert j.n == 0, "the riding step does its own bookkeeping"
    e = _env_step_job(step)
    check(self.L.dq_qnet_td_backward_adam_env(self._h, ptr(params), ctypes.byref(j), ptr(grads), ptr(m), ptr(v), float(lr), float(beta_1),
                                              float(beta_2), float(epsilon), int(t), env_handle, ctypes.byref(e), self._stream()))
    return grads


def _td_backward_phase0_env(self, params, td, grads, env_handle, step):
    """The several-GPU form of td_backward_adam_env: td_backward_phase0() carrying the environment step."""
    j = _td_job(td)
    assert j.n == 0, "the riding step does its own bookkeeping"
    e = _env_step_job(step)
    check(self.L.dq_qnet_td_backward_phase0_env(self._h, ptr(params), ctypes.byref(j), ptr(grads), env_handle, ctypes.byref(e), self._stream()))
    return grads


QNetwork.td_backward_phase0_env = _td_backward_phase0_env
QNetwork.td_backward_adam_env = _td_backward_adam_env


def _net_adam_step(self, params, grads, m, v, t, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    """adam_step() whose skipped (non-finite) gradient elements raise this network's range flag (dq_qnet_adam_step): the several-GPU
    branch's optimizer step behind the gradient all-reduce."""
    check(self.L.dq_qnet_adam_step(self._h, ptr(params), ptr(grads), ptr(m), ptr(v), float(lr), float(beta_1), float(beta_2), float(epsilon),
                                   int(t), self._stream()))


QNetwork.adam_step = _net_adam_step


def _td_backward_phase0(self, params, td, grads):
    """The several-GPU form: TD step + phase 0 (dueling + dense layers) of the backward; backward_phase(.., 1) follows."""
    j = _td_job(td)
    check(self.L.dq_qnet_td_backward_phase0(self._h, ptr(params), ctypes.byref(j), ptr(grads), self._stream()))
    return grads


QNetwork.td_backward_phase0 = _td_backward_phase0
QNetwork.td_backward_adam = _td_backward_adam


def td_target(q_online_s1, q_target_s1, reward, terminal, gamma, index=None, out=None):
    B, A = q_online_s1.shape
    if out is None:
        out = torch.empty(B, dtype=torch.float32, device=q_online_s1.device)
    check(_lib.lib().dq_td_target(ptr(q_online_s1), ptr(q_target_s1), ptr(reward), ptr(terminal), ptr(index), float(gamma), B, A,
                                  ptr(out), _lib.current_stream(q_online_s1.device)))
    return out


def td_loss_grad(q_s0, action, y, grad_scale=None, index=None, dq=None, metrics=None):
    B, A = q_s0.shape
    if dq is None:
        dq = torch.empty_like(q_s0)
    if metrics is None:
        metrics = torch.empty(TD_METRICS_FLOATS, dtype=torch.float32, device=q_s0.device)
    check(_lib.lib().dq_td_loss_grad(ptr(q_s0), ptr(action), ptr(index), ptr(y), B, A, 1.0 / B if grad_scale is None else float(grad_scale),
                                     ptr(dq), ptr(metrics), _lib.current_stream(q_s0.device)))
    return dq, metrics


def td_update(q_online_s1, q_target_s1, q_s0, reward, terminal, action, gamma, grad_scale=None, index=None, y=None, dq=None, metrics=None,
              step_stats=None):
    """td_target + td_loss_grad in one launch (dq_td_update); metrics[0..1] are valid only after td_metrics().
    step_stats = (done, was_reset, lifetime, reward, n, stats): the episode bookkeeping of the step just taken rides along
    (dq_td_update_stats)."""
    B, A = q_s0.shape
    if dq is None:
        dq = torch.empty_like(q_s0)
    if step_stats is not None:
        done, was_reset, lifetime, step_reward, n, stats = step_stats
        check(_lib.lib().dq_td_update_stats(ptr(q_online_s1), ptr(q_target_s1), ptr(q_s0), ptr(reward), ptr(terminal), ptr(action), ptr(index),
                                            float(gamma), B, A, 1.0 / B if grad_scale is None else float(grad_scale), ptr(y), ptr(dq),
                                            ptr(metrics), ptr(done), ptr(was_reset), ptr(lifetime), ptr(step_reward), int(n), ptr(stats),
                                            _lib.current_stream(q_s0.device)))
        return dq
    check(_lib.lib().dq_td_update(ptr(q_online_s1), ptr(q_target_s1), ptr(q_s0), ptr(reward), ptr(terminal), ptr(action), ptr(index), float(gamma),
                                  B, A, 1.0 / B if grad_scale is None else float(grad_scale), ptr(y), ptr(dq), ptr(metrics),
                                  _lib.current_stream(q_s0.device)))
    return dq


def td_metrics(metrics, batch):
    """Final fixed-order reduction of the per-block loss / mean_q partials into metrics[0], metrics[1]."""
    check(_lib.lib().dq_td_metrics(ptr(metrics), int(batch), _lib.current_stream(metrics.device)))
    return metrics


def adam_step(params, grads, m, v, t, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
    check(_lib.lib().dq_adam_step(ptr(params), ptr(grads), ptr(m), ptr(v), params.numel(), float(lr), float(beta_1), float(beta_2),
                                  float(epsilon), int(t), _lib.current_stream(params.device)))


def replay_sample_multi(terminal_ring, n_envs, n_slots, head_slot, filled_slots, batch, seed, t0, n_updates, sample_base=0, out=None):
    """index[u] = replay_sample(..., t = t0 + u): the minibatches of n_updates consecutive updates on one ring state, one launch."""
    if out is None:
        out = torch.empty((n_updates, batch), dtype=torch.int32, device=terminal_ring.device)
    assert out.is_contiguous() and out.numel() >= n_updates * batch
    check(_lib.lib().dq_replay_sample_multi(ptr(terminal_ring), int(n_envs), int(n_slots), int(head_slot), int(filled_slots), int(batch),
                                            _seed_arr(seed), int(t0), int(n_updates), int(sample_base), ptr(out), _lib.current_stream(terminal_ring.device)))
    return out


def replay_sample(terminal_ring, n_envs, n_slots, head_slot, filled_slots, batch, seed, t, sample_base=0, out=None):
    if out is None:
        out = torch.empty(batch, dtype=torch.int32, device=terminal_ring.device)
    check(_lib.lib().dq_replay_sample(ptr(terminal_ring), int(n_envs), int(n_slots), int(head_slot), int(filled_slots), int(batch),
                                      _seed_arr(seed), int(t), int(sample_base), ptr(out), _lib.current_stream(terminal_ring.device)))
    return out
