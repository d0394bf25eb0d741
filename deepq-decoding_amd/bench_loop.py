"""Full hot-path loop used by bench.py: per vector step every lattice acts (Q forward + epsilon-greedy over
legal moves), the environment kernel steps them into the replay ring, and one DQN minibatch update runs."""
import ctypes
import os

import torch

from . import _lib

from .core import DQNCore
from .env import VectorEnv
from .qnet import QNetwork

# the reference's network (fixed_config: c_layers, ff_layers, dueling=True; Generate_Base_Configs...py:18-19,24)
C_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]]
FF_LAYERS = [[512, 0.2]]
MFMA_F32_PEAK_TFLOPS = 157.3
MFMA_F16_PEAK_TFLOPS = 2500.0        # dense f16 / bf16 (MI355X_MICROARCH.md); the fused chains issue 2 (binary operand) or 3 f16 MFMAs per f32 product
HBM_PEAK_GBS = 8000.0


from ._digest import code_only as _code_only, csrc_digest      # (the stamp of a PMC pass = the code digest of the kernel sources it measured)


def pmc_record(mode="loop", config="c3"):
    """The committed PMC pass of (mode, config) as a dict, or None when the file is missing / unreadable."""
    import json
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"pmc_traffic_{mode}_{config}.json")
    try:
        with open(path) as f:
            return json.load(f)
    except (OSError, ValueError):
        return None


def pmc_traffic(symbol, mode="loop", config="c3", minibatch=0, updates_per_step=1, lattices=0):
    """HBM bytes per launch of the kernel SYMBOL that ran (dq_prof_kernel_symbol: `conv_wave_kernel`, not its family alias) from the committed PMC pass
    profiles/pmc_traffic_<mode>_<config>.json (tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction
    applied).  PMC counters cannot be collected inside a timed run, so this is a recorded value -- valid only for the kernel code it was taken with AND
    for the launch shape it was taken at: the file carries the sources' code digest (csrc_digest) and the pass's minibatch / updates per step /
    lattices (0 = the configuration's default); anything else (stale pass, another shape, another form of the family, missing file) gives None."""
    rec = pmc_record(mode, config)
    try:
        if rec is None or rec.get("csrc_sha256") != csrc_digest():
            return None
        shape = rec.get("shape", {"minibatch": 0, "updates_per_step": 1, "lattices": 0})
        if (int(shape.get("minibatch", 0)), int(shape.get("updates_per_step", 1)), int(shape.get("lattices", 0))) != (int(minibatch), int(updates_per_step), int(lattices)):
            return None
        return rec["kernels"][symbol]["hbm_bytes_per_launch_corrected"]
    except (KeyError, ValueError, TypeError):
        return None


def prof_stride(steps):
    """A timed launch costs its stream a few microseconds (measured: the c3 loop 150.1 us per step with no launch timed, 154 with the dominant
    kernel's launch timed in every step; the acting mode 41.4 against 45.8), so only a sample of the armed family's launches is timed: about 64 over
    the timed region, every launch when the region is that short."""
    samples = 64 if steps >= 256 else max(5, steps // 4)
    return max(1, steps // samples)


def prof_arm(L, family, steps, per_step):
    """Arms `family` for the timed region of `steps` steps (per_step launches each); False if the library has no such family."""
    for i in range(L.dq_prof_kernel_count()):
        if L.dq_prof_kernel_name(i).decode() == family:
            stride = prof_stride(steps)
            _lib.check(L.dq_prof_arm(i, steps * per_step // stride + 8))
            _lib.check(L.dq_prof_stride(stride))
            return True
    return False


def prof_collect(L):
    n, ms = ctypes.c_int(), ctypes.c_double()
    _lib.check(L.dq_prof_collect(ctypes.byref(n), ctypes.byref(ms)))
    _lib.check(L.dq_prof_arm(-1, 0))
    return n.value, ms.value


class FullLoop:
    # f32 results (1e-5 against the float64 oracle); the hot products are issued on the f16 matrix pipe, each operand as two f16 pieces
    dtype = "f32 (operands as 2 f16 pieces = 22 significant bits, 2 or 3 f16 MFMAs per product, f32 accumulate)"

    def __init__(self, dq, cfg, n_local, rank, world, minibatch, eps=0.1, replay_transitions=1 << 20, lr=1e-4,
                 target_every=32, mode="loop", config_name="c3", updates_per_step=1):
        self.cfg, self.n, self.rank, self.world, self.B = cfg, n_local, rank, world, minibatch
        self.k = int(updates_per_step)           # minibatch updates per vector step (DQNAgent.updates_per_vector_step): the reference's replay ratio of
                                                 # 32 trained samples per environment step (TRAIN:119-127) is k * B / n = 32
        self.eps, self.target_every, self.mode, self.config_name = eps, target_every, mode, config_name
        self.unit = "dqn_samples/s" if mode == "learn" else "env_steps/s"
        self.env = VectorEnv(n_envs=n_local, env_id_base=rank * n_local, **cfg)
        self.net = QNetwork(self.env.obs_shape, C_LAYERS, FF_LAYERS, self.env.num_actions, dueling=True, max_batch=max(n_local, minibatch))
        self.core = DQNCore(self.env, self.net, batch_size=minibatch, memory_limit=replay_transitions, gamma=0.99, lr=lr,
                            rank=rank, world_size=world)
        self.core.reset_env()
        # untimed set-up: the replay ring is FILLED by acting (keras-rl's nb_steps_warmup phase: transitions are collected before learning
        # starts), so that every update of the warm-up and of the timed region samples a full memory -- 2^20 transitions spread over the whole
        # 0.9 GB ring, as in training -- instead of the handful of slots a cold start holds
        for _ in range(max(4, self.core.T - 1)):
            self.core.act_and_step(self.eps)
        self.layer_macs = self._layer_macs()
        self.macs = sum(self.layer_macs)
        self.L = _lib.lib()
        self.prof_family = None
        self.pmc_minibatch = self.pmc_lattices = 0    # what bench.py was ASKED for (0 = the configuration's default): the key of the recorded PMC pass (pmc_traffic)

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
        k = self.k if self.mode == "loop" else 1
        fwd_samples = {"loop": self.n + 3 * self.B * k, "act": self.n, "learn": 3 * self.B}[self.mode]
        B = self.B * k
        if self.mode == "act":
            assert self.net.fused_supported
            return {"conv_chain_kernel": (1, 2.0 * conv * fwd_samples, "mfma"), "dense_chain_kernel": (1, 2.0 * dense * fwd_samples, "mfma")}
        if self.net.fused_supported:
            # one forward launch pair per step (the acting forward and the update's three forwards share it), one launch per
            # backward kernel
            return {
                "conv_chain_kernel": (k, 2.0 * conv * fwd_samples, "mfma"),
                "dense_chain_kernel": (k, 2.0 * dense * fwd_samples, "mfma"),
                # conv weight gradients (= forward MACs) + data gradients through conv3 and conv2
                "conv_bwd_chain_kernel": (k, 2.0 * (conv + sum(lm[1:nc])) * B, "mfma"),
                "dense_bwd_chain_kernel": (k, 2.0 * dense * B, "mfma"),
                "dense_wgrad_kernel": (k, 2.0 * dense * B, "mfma"),
            }
        return {
            "gemm_fwd_kernel": (5 * len(lm) - 1, 2.0 * self.macs * fwd_samples + 2.0 * (self.macs - lm[0]) * self.B, "mfma"),
            "gemm_wgrad_kernel": (len(lm), 2.0 * self.macs * self.B, "mfma"),
        }

    def f16_pipe_factor(self):
        """family -> f16 MFMA flops issued per algorithmic f32 flop (work-weighted): every fused kernel runs on the f16 pipe (f16x2,
        csrc/qnet.h) -- 3 MFMAs per product, 2 where one operand is the binary observation (conv1 forward, its weight gradient)."""
        nc, lm = len(C_LAYERS), self.layer_macs
        if not self.net.fused_supported:
            return {}
        conv, rest = sum(lm[:nc]), sum(lm[1:nc])
        # the first convolution's operand is binary: 2 MFMAs per product.  On the uint8 image its K = k*k*C is padded to whole blocks of 32 and every
        # cell is multiplied, constants included; on patch words (DQNCore.compact) only the K_data <= 32 data cells are: ONE block of 32 in the
        # forward, K_data + 5 columns (padded to 16s) in its weight gradient -- fewer ISSUED flops for the same algorithmic work
        c1_fwd = c1_bwd = 2.0 * lm[0]
        if getattr(self.core, "compact", False):
            K1 = C_LAYERS[0][1] ** 2 * self.env.obs_shape[0]
            kd = 4 * self.env.volume_depth + self.env.n_action_layers
            c1_fwd = 2.0 * lm[0] * 32.0 / K1
            c1_bwd = 2.0 * lm[0] * (16.0 * ((kd + 5 + 15) // 16)) / K1
        conv_fwd = (c1_fwd + 3.0 * rest) / conv
        d = self.env.d
        if getattr(self.core, "compact", False) and d == 5 and 4 * self.env.volume_depth + self.env.n_action_layers + 6 <= 32 \
                and os.environ.get("DQ_CONV_FORM", "w")[:1] != "g":
            # the wave-private form (csrc/conv_wave.hip): whole 16-row tiles per sample -- 2 tiles for conv1's 25 pixels, one for conv2's 16 and for
            # conv3's 9 -- so the ISSUED count includes the padding rows: 2 x 32 x 64 x 32 + 3 x 16 x 32 x 256 + 3 x 16 x 32 x 128 MACs per sample
            conv_fwd = (2.0 * 32 * 64 * 32 + 3.0 * 16 * 32 * 256 + 3.0 * 16 * 32 * 128) / conv
        conv_bwd = (c1_bwd + 3.0 * rest + 3.0 * rest) / (conv + rest)      # weight gradients + data gradients
        Bk = self.B
        if getattr(self.core, "compact", False) and d == 5 and 4 * self.env.volume_depth + self.env.n_action_layers + 6 <= 32 and Bk % 8 == 0 and Bk >= 1024 \
                and os.environ.get("DQ_CONV_BWD_FORM", "16")[:1] != "8":
            # the 16-wave form (csrc/conv_bwd16.hip), per sample of a group of 8: g2 16 rows x 128 x 32, g1 26 rows (13 tiles of 16 for 200 pixels) x 128 x 64,
            # dW3 12 rows (three blocks of 32 for 72) x 128 x 32, dW2 16 x 256 x 32 -- three MFMAs per product -- and dW1 32 rows (eight blocks for 200) x 32 x 64, two
            conv_bwd = (3.0 * (16 * 128 * 32 + 26 * 128 * 64 + 12 * 128 * 32 + 16 * 256 * 32) + 2.0 * 32 * 32 * 64) / (conv + rest)
            if os.environ.get("DQ_CONV_FORM", "w")[:1] != "g" and os.environ.get("DQ_CONV_BWD_A1", "r")[:1] != "s":
                # round 6: a1 recomputed from the patch words (26 rows per sample x K 32 x 64 channels, one MFMA per weight piece) instead of read back from HBM
                conv_bwd += 2.0 * 26 * 32 * 64 / (conv + rest)
        return {"conv_chain_kernel": conv_fwd,
                "conv_bwd_chain_kernel": conv_bwd,
                "dense_chain_kernel": 3.0, "dense_bwd_chain_kernel": 3.0, "dense_wgrad_kernel": 3.0}

    def _family_id(self, name):
        for i in range(self.L.dq_prof_kernel_count()):
            if self.L.dq_prof_kernel_name(i).decode() == name:
                return i
        raise KeyError(name)

    def _collect(self):
        n, ms, lo, hi = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        _lib.check(self.L.dq_prof_collect_spread(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(lo), ctypes.byref(hi)))
        self.last_spread_ms = (lo.value, hi.value)
        return n.value, ms.value

    def reference_ratio_leg(self, steps=100, warmup=10):
        """A second timed region at the REFERENCE's replay ratio -- 32 trained samples per environment step (keras-rl train_interval = 1, batch_size = 32:
        TRAIN:119-127), i.e. k = 32 n / B minibatch updates of B per vector step of n lattices -- on the same loop, ring and parameters: `warmup` untimed
        vector steps, then `steps` timed ones between two device synchronisations.  One GPU only (bench.py runs it behind the headline region)."""
        import time
        k_old, self.k = self.k, max(1, 32 * self.n // self.B)
        try:
            for _ in range(warmup):
                self.step(timed=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                self.step(timed=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            k = self.k
        finally:
            self.k = k_old
        return dict(value=self.n * steps / dt, unit="env_steps/s", ms_per_step=1e3 * dt / steps, steps=steps, warmup=warmup, updates_per_vector_step=k,
                    minibatch=self.B, samples_trained_per_env_step=k * self.B / self.n, updates_per_s=k * steps / dt, dqn_samples_per_s=k * steps * self.B / dt,
                    us_per_update=1e6 * dt / (steps * k),
                    reference="one 32-sample minibatch per environment step: cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:119-127")

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

    def family_times(self, steps=200):
        """Diagnostic (DQ_BENCH_FAMILIES=1): every family's average launch duration in the free-running loop, one family timed at a time."""
        out = {}
        for name, (per_step, _, _) in self.kernel_families().items():
            _lib.check(self.L.dq_prof_arm(self._family_id(name), steps * per_step + 8))
            for _ in range(steps):
                self.step(timed=False)
            n, ms = self._collect()
            if n:
                out[name] = dict(launches=n, avg_us=ms * 1e3 / n, per_step=per_step)
        _lib.check(self.L.dq_prof_arm(-1, 0))
        return out

    def arm(self, steps):
        if self.prof_family:
            prof_arm(self.L, self.prof_family, steps, self.kernel_families()[self.prof_family][0])

    def units_per_step(self):
        return self.B if self.mode == "learn" else self.n

    def step(self, timed):
        if self.mode == "act":
            self.core.act_and_step(self.eps)
            return
        if self.mode == "learn":
            self.core.update()
        else:
            self.core.step_and_update(self.eps, extra_updates=self.k - 1)      # == act_and_step() + k update(), the first update's forwards with the acting forward
        if self.core.updates % self.target_every < (self.k if self.mode == "loop" else 1):
            self.core.update_target_hard()

    def config(self):
        return dict(policy=f"eps-greedy over legal moves, eps={self.eps}, live Q-network", minibatch_per_gpu=self.B,
                    updates_per_vector_step=self.k if self.mode == "loop" else (0 if self.mode == "act" else 1),
                    replay_ratio_samples_per_env_step=(self.k * self.B / self.n) if self.mode == "loop" else None,
                    reference_replay_ratio_samples_per_env_step=32, observations="patch words (d*d u32 per lattice)" if self.core.compact else "uint8 planes",
                    replay_ring_slots=self.core.T, network="conv 64x3s2-32x2-32x2, dense 512 (dropout .2), dueling",
                    n_params=self.net.n_params, grad_allreduce="RCCL sum of the flat fp32 gradient" if self.world > 1 else "none (1 GPU)")

    def report(self, steps, dt, world):
        ep, life, rew, stepped = self.core.read_stats()
        # acting forward + update (3 fwd + bwd = 5x fwd)
        flops_per_step = 2 * self.macs * {"loop": self.n + self.B * 5 * self.k, "act": self.n, "learn": self.B * 5}[self.mode]
        roof = None
        if self.prof_family:
            launches, ms = self._collect()
            _lib.check(self.L.dq_prof_arm(-1, 0))
            per_step, work, bound = self.kernel_families()[self.prof_family]
            if launches:
                avg_s = ms * 1e-3 / launches
                per_launch = work / per_step
                achieved = per_launch / avg_s / 1e12
                issued = self.f16_pipe_factor().get(self.prof_family)
                # the peak is that of the pipe the kernel runs on: f32-class products cost `issued` f16 MFMA flops each, so the f16
                # pipe's dense peak divided by that (frac = issued f16 flops / f16 peak); the f32-input MFMA peak is quoted beside it
                peak = MFMA_F16_PEAK_TFLOPS / issued if issued else MFMA_F32_PEAK_TFLOPS
                # the symbol the family's launches used (conv_chain_kernel is a family: conv_wave_kernel / conv_chain_pkernel / conv_chain_kernel)
                symbol = self.L.dq_prof_kernel_symbol(self._family_id(self.prof_family)).decode() or self.prof_family
                roof = dict(kernel=symbol, family=self.prof_family, bound=bound, achieved=achieved, peak=peak, unit="TFLOP/s",
                            frac=achieved / peak, traffic=pmc_traffic(symbol, self.mode, self.config_name, self.pmc_minibatch, self.k, self.pmc_lattices), avg_launch_us=avg_s * 1e6,
                            min_launch_us=self.last_spread_ms[0] * 1e3, max_launch_us=self.last_spread_ms[1] * 1e3,
                            launches_timed=launches, algorithmic_flops_per_launch=per_launch)
                if issued:
                    roof["pipe"] = dict(name="f16 MFMA (v_mfma_f32_16x16x32_f16)", peak=MFMA_F16_PEAK_TFLOPS, issued_tflops=achieved * issued,
                                        mfmas_per_f32_product=issued)
                    roof["vs_f32_mfma_peak"] = dict(peak=MFMA_F32_PEAK_TFLOPS, ratio=achieved / MFMA_F32_PEAK_TFLOPS)
        out = {
            "roofline": roof,
            "dqn_updates_per_s": 0.0 if self.mode == "act" else steps * (self.k if self.mode == "loop" else 1) / dt,
            "dqn_samples_per_s": 0.0 if self.mode == "act" else steps * (self.k if self.mode == "loop" else 1) * self.B * world / dt,
            "achieved_qnet_tflops_per_gpu": flops_per_step * steps / dt / 1e12,
            "episodes_finished_rank0": ep,
            "mean_lifetime_rank0": (life / ep) if ep else None,
        }
        return out
