"""Full hot-path loop used by bench.py: per vector step every lattice acts (Q forward + epsilon-greedy over
legal moves), the environment kernel steps them into the replay ring, and one DQN minibatch update runs."""
import time

import torch

from .core import DQNCore
from .env import VectorEnv
from .qnet import QNetwork

# the reference's network (fixed_config: c_layers, ff_layers, dueling=True; Generate_Base_Configs...py:18-19,24)
C_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]]
FF_LAYERS = [[512, 0.2]]
MFMA_F32_PEAK_TFLOPS = 157.3
HBM_PEAK_GBS = 8000.0


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
        self.macs = self._forward_macs()

    def _forward_macs(self):
        c, h, w = self.env.obs_shape
        macs = 0
        for f, k, s in C_LAYERS:
            oh, ow = (h - k) // s + 1, (w - k) // s + 1
            macs += oh * ow * f * k * k * c
            c, h, w = f, oh, ow
        n = c * h * w
        for u, _ in FF_LAYERS:
            macs += n * u
            n = u
        A = self.env.num_actions
        return macs + n * A + A * (A + 1)

    def step(self, timed):
        self.core.act_and_step(self.eps)
        self.core.update()
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
        out = {
            "dqn_updates_per_s": steps / dt,
            "dqn_samples_per_s": steps * self.B * world / dt,
            "achieved_qnet_tflops_per_gpu": flops_per_step * steps / dt / 1e12,
            "episodes_finished_rank0": ep,
            "mean_lifetime_rank0": (life / ep) if ep else None,
        }
        return out

    def cpu_baseline(self, cfg):
        from bench import cpu_baseline_env          # env half on the C oracle
        base = cpu_baseline_env(cfg, seconds=8.0)
        return base
