"""Full hot-path loop used by bench.py: per vector step every lattice acts (Q forward + epsilon-greedy over
legal moves), the environment kernel steps them into the replay ring, and one DQN minibatch update runs."""
import ctypes

import torch

from . import _lib

from .core import DQNCore
from .env import VectorEnv
from .qnet import QNetwork

# the reference's network (fixed_config: c_layers, ff_layers, dueling=True; Generate_Base_Configs...py:18-19,24)
C_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]]
FF_LAYERS = [[512, 0.2]]
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_BF16_PEAK_TFLOPS = 2500.0       # dense bf16 (MI355X_MICROARCH.md); the fused chains issue 3 (binary operand) or 6 bf16 MFMAs per f32 product
HBM_PEAK_GBS = 8000.0


def _pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/r01_pmc_traffic.json, produced by
    tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction applied); None if the
    file or the kernel is missing.  PMC counters cannot be collected inside a timed run, so this is the last recorded value."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_pmc_traffic.json")
    try:
        with open(path) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch_corrected"]
    except (OSError, KeyError, ValueError):
        return None


class FullLoop:
    dtype = "f32"

    def __init__(self, dq, cfg, n_local, rank, world, minibatch, eps=0.1, replay_transitions=1 << 20, lr=1e-4,
                 target_every=32):
        self.cfg, self.n, self.rank, self.world, self.B = cfg, n_local, rank, world, minibatch
        self.eps, self.target_every = eps, target_every
        self.env = VectorEnv(n_envs=n_local, env_id_base=rank * n_local, **cfg)
        self.net = QNetwork(self.env.obs_shape, C_LAYERS, FF_LAYERS, self.env.num_actions, dueling=True, max_batch=max(n_local, minibatch))
        self.core = DQNCore(self.env, self.net, batch_size=minibatch, memory_limit=replay_transitions, gamma=0.99, lr=lr,
                            rank=rank, world_size=world)
        self.core.reset_env()
        for _ in range(4):                      # a few transitions before the first update
            self.core.act_and_step(self.eps)
        self.layer_macs = self._layer_macs()
        self.macs = sum(self.layer_macs)
        self.L = _lib.lib()
        self.prof_family = None

    def _layer_macs(self):
        """Multiply-accumulates per sample of each layer (conv1 100800, conv2 131072, conv3 36864, dense 147456,
        26112, 2652 at c3)."""
        c, h, w = self.env.obs_shape
        macs = []
        for f, k, s in C_LAYERS:
            oh, ow = (h - k) // s + 1, (w - k) // s + 1
            macs.append(oh * ow * f * k * k * c)
            c, h, w = f, oh, ow
        n = c * h * w
        for u, _ in FF_LAYERS:
            macs.append(n * u)
            n = u
        A = self.env.num_actions
        return macs + [n * A, A * (A + 1)]

    # -- live roofline of the dominant kernel family (HIP events inside the library, on the launch stream) -----------------
    def kernel_families(self):
        """family name -> (launches per vector step, algorithmic flops or bytes per vector step, bound).  One vector step =
        acting forward on n lattices + three forwards and one backward on the minibatch."""
        nc = len(C_LAYERS)
        lm = self.layer_macs
        conv, dense = sum(lm[:nc]), sum(lm[nc:])
        fwd_samples = self.n + 3 * self.B
        if self.net.fused_supported:
            # one forward launch pair per step (the acting forward and the update's three forwards share it), one launch per
            # backward kernel
            return {
                "conv_chain_kernel": (1, 2.0 * conv * fwd_samples, "mfma"),
                "dense_chain_kernel": (1, 2.0 * dense * fwd_samples, "mfma"),
                # conv weight gradients (= forward MACs) + data gradients through conv3 and conv2
                "conv_bwd_chain_kernel": (1, 2.0 * (conv + sum(lm[1:nc])) * self.B, "mfma"),
                "dense_bwd_chain_kernel": (1, 2.0 * dense * self.B, "mfma"),
                "dense_wgrad_kernel": (1, 2.0 * dense * self.B, "mfma"),
            }
        return {
            "gemm_fwd_kernel": (5 * len(lm) - 1, 2.0 * self.macs * fwd_samples + 2.0 * (self.macs - lm[0]) * self.B, "mfma"),
            "gemm_wgrad_kernel": (len(lm), 2.0 * self.macs * self.B, "mfma"),
        }

    def bf16_pipe_factor(self):
        """family -> bf16 MFMA flops issued per algorithmic f32 flop (work-weighted) for the kernels that run on the bf16 pipe."""
        nc, lm = len(C_LAYERS), self.layer_macs
        if not self.net.fused_supported:
            return {}
        conv = sum(lm[:nc])
        return {"conv_chain_kernel": (3.0 * lm[0] + 6.0 * sum(lm[1:nc])) / conv}      # conv1's operand is binary: 3 pieces suffice

    def _family_id(self, name):
        for i in range(self.L.dq_prof_kernel_count()):
            if self.L.dq_prof_kernel_name(i).decode() == name:
                return i
        raise KeyError(name)

    def _collect(self):
        n, ms = ctypes.c_int(), ctypes.c_double()
        _lib.check(self.L.dq_prof_collect(ctypes.byref(n), ctypes.byref(ms)))
        return n.value, ms.value

    def pick_dominant(self, probe_steps=4):
        """Times every family over a few untimed steps and arms the one with the largest total duration."""
        best, best_ms = None, -1.0
        for name, (per_step, _, _) in self.kernel_families().items():
            _lib.check(self.L.dq_prof_arm(self._family_id(name), probe_steps * per_step + 8))
            for _ in range(probe_steps):
                self.step(timed=False)
            n, ms = self._collect()
            if n and ms > best_ms:
                best, best_ms = name, ms
        self.prof_family = best
        _lib.check(self.L.dq_prof_arm(-1, 0))
        return best

    def arm(self, steps):
        if self.prof_family:
            per_step = self.kernel_families()[self.prof_family][0]
            _lib.check(self.L.dq_prof_arm(self._family_id(self.prof_family), steps * per_step + 8))

    def step(self, timed):
        self.core.step_and_update(self.eps)          # == act_and_step() + update(), the four forwards in one pair of launches
        if self.core.updates % self.target_every == 0:
            self.core.update_target_hard()

    def config(self):
        return dict(policy=f"eps-greedy over legal moves, eps={self.eps}, live Q-network", minibatch_per_gpu=self.B,
                    updates_per_vector_step=1, replay_ratio_samples_per_env_step=self.B / self.n,
                    replay_ring_slots=self.core.T, network="conv 64x3s2-32x2-32x2, dense 512 (dropout .2), dueling",
                    n_params=self.net.n_params, grad_allreduce="RCCL sum of the flat fp32 gradient" if self.world > 1 else "none (1 GPU)")

    def report(self, steps, dt, world):
        ep, life, rew, stepped = self.core.read_stats()
        flops_per_step = 2 * self.macs * (self.n * 1 + self.B * 5)       # acting forward + update (3 fwd + bwd = 5x fwd)
        roof = None
        if self.prof_family:
            launches, ms = self._collect()
            _lib.check(self.L.dq_prof_arm(-1, 0))
            per_step, work, bound = self.kernel_families()[self.prof_family]
            if launches:
                avg_s = ms * 1e-3 / launches
                per_launch = work / per_step
                achieved = per_launch / avg_s / 1e12
                roof = dict(kernel=self.prof_family, bound=bound, achieved=achieved, peak=MFMA_F32_PEAK_TFLOPS, unit="TFLOP/s",
                            frac=achieved / MFMA_F32_PEAK_TFLOPS, traffic=_pmc_traffic(self.prof_family), avg_launch_us=avg_s * 1e6,
                            launches_timed=launches, algorithmic_flops_per_launch=per_launch)
                issued = self.bf16_pipe_factor().get(self.prof_family)
                if issued:
                    # the same launch seen from the pipe it actually runs on: f32-accurate products issued as bf16 MFMAs
                    roof["bf16_pipe"] = dict(issued_tflops=achieved * issued, peak=MFMA_BF16_PEAK_TFLOPS,
                                             frac=achieved * issued / MFMA_BF16_PEAK_TFLOPS, mfmas_per_f32_product=issued)
        out = {
            "roofline": roof,
            "dqn_updates_per_s": steps / dt,
            "dqn_samples_per_s": steps * self.B * world / dt,
            "achieved_qnet_tflops_per_gpu": flops_per_step * steps / dt / 1e12,
            "episodes_finished_rank0": ep,
            "mean_lifetime_rank0": (life / ep) if ep else None,
        }
        return out

    def cpu_baseline(self, cfg, seconds=15.0):
        """The SAME loop on this host's cores, assembled from the oracles (test infrastructure, used here only as the timed
        CPU baseline): C oracle environment (a port of Environments.py) + numpy float64 restatement of the keras-rl / Keras
        update (BLAS-threaded im2col GEMMs).  Bounded: whole vector steps until ~`seconds` have elapsed (at least 2)."""
        import os
        import time

        import numpy as np

        from oracle import c_oracle, dqn_oracle as O
        n, B, eps, gamma = cfg["n_envs"], self.B, self.eps, 0.99
        kw = {k: v for k, v in cfg.items() if k != "n_envs"}
        env = c_oracle.COracleEnv(n_envs=n, **kw)
        spec = O.QNetSpec(env.obs_shape, C_LAYERS, FF_LAYERS, env.num_actions)
        flat = O.glorot_init(spec, (1, 2)).astype(np.float64)
        target, m, v = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
        T = 6
        ring_obs = np.zeros((T, n) + env.obs_shape, np.uint8)
        ring_a, ring_r, ring_t = np.zeros((T, n), np.int32), np.zeros((T, n), np.float32), np.zeros((T, n), np.uint8)
        rng = np.random.RandomState(0)
        ring_obs[0] = env.reset()
        cur, filled, steps, t0 = 0, 1, 0, time.perf_counter()
        while steps < 2 or time.perf_counter() - t0 < seconds:
            q, _ = O.forward(spec, flat, ring_obs[cur])
            a = np.where(rng.rand(n) < eps, env.policy_uniform_legal(steps), q.argmax(axis=1)).astype(np.int32)
            nxt = (cur + 1) % T
            obs, r, done = env.step(a, auto_reset=True)
            ring_obs[nxt], ring_a[cur], ring_r[cur], ring_t[cur] = obs, a, r, done
            cur, filled, steps = nxt, min(T, filled + 1), steps + 1
            back = rng.randint(1, filled, size=B)                     # transition = (slot cur-back, slot cur-back+1)
            e = rng.randint(0, n, size=B)
            s0 = (cur - back) % T
            s1 = (s0 + 1) % T
            q1_t, _ = O.forward(spec, target, ring_obs[s1, e])
            q1_o, _ = O.forward(spec, flat, ring_obs[s1, e])
            y = O.td_targets(q1_o, q1_t, ring_r[s0, e], ring_t[s0, e], gamma)
            keep = rng.rand(B, FF_LAYERS[0][0]) >= FF_LAYERS[0][1]
            q0, cache = O.forward(spec, flat, ring_obs[s0, e], training=True, keep_masks=[keep])
            _, _, dq = O.loss_and_grad(q0, ring_a[s0, e], y)
            g = O.backward(spec, flat, cache, dq)
            flat, m, v = O.adam_step(flat, g, m, v, steps, 1e-4)
        dt = time.perf_counter() - t0
        return dict(value=n * steps / dt, unit="env_steps/s", cores=os.cpu_count(), kind="port",
                    sample=f"{steps} vector steps x {n} lattices (acting forward + C-oracle env step + one {B}-sample double-DQN update "
                           f"in numpy float64, BLAS threads = all cores), {dt:.1f}s",
                    dqn_updates_per_s=steps / dt)
