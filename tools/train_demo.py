"""Trains a decoder from scratch on the GPU path through the reference's own API (build_convolutional_nn, DQNAgent.fit / test)
and prints the greedy lifetimes -- a behavioural check that the whole loop learns (GPU box).

    python tools/train_demo.py [config=c2] [vector_steps=20000] [lr=3e-4]
"""
import importlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dq = importlib.import_module("deepq-decoding_amd")
ag = importlib.import_module("deepq-decoding_amd.agent")

CONFIGS = {
    "c2": dict(d=5, error_model="X", use_Y=False, volume_depth=5, p_phys=0.007, p_meas=0.007),
    "c3": dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011),
}
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
vsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 3e-4
N = 4096
cfg = CONFIGS[name]
env = dq.VectorEnv(n_envs=N, **cfg)
model = ag.build_convolutional_nn([[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], env.obs_shape, env.num_actions)
memory = ag.SequentialMemory(limit=1 << 20, window_length=1)
policy = ag.LinearAnnealedPolicy(ag.EpsGreedyQPolicy(masked_greedy=False), attr="eps", value_max=1.0, value_min=0.02, value_test=0.0,
                                 nb_steps=N * vsteps // 4)
dqn = ag.DQNAgent(model=model, nb_actions=env.num_actions, memory=memory, nb_steps_warmup=N * 8, target_model_update=N * 250,
                  policy=policy, test_policy=ag.GreedyQPolicy(masked_greedy=True), gamma=0.99, enable_dueling_network=True,
                  batch_size=N)
dqn.compile(ag.Adam(lr=lr))
t0 = time.time()
hist = dqn.fit(env, nb_steps=N * vsteps, verbose=0, log_interval=N * 1000, episode_averaging_length=2000, success_threshold=None,
               stopping_patience=None, min_nb_steps=N * vsteps, single_cycle=False, sync_interval=500)
dt = time.time() - t0
h = hist.history
print(f"trained {N * vsteps / 1e6:.0f} M env steps in {dt:.1f} s ({N * vsteps / dt / 1e6:.2f} M steps/s incl. host logging)")
for k in ("episode_lifetimes_rolling_avg", "loss", "mean_q"):
    if k in h and len(h[k]):
        v = np.asarray(h[k], dtype=float)
        print(k, "first/mid/last:", v[0], v[len(v) // 2], v[-1])
for p in (cfg["p_phys"], cfg["p_phys"] * 0.5):
    env.p_phys = env.p_meas = p
    th = dqn.test(env, nb_episodes=2048, visualize=False, verbose=0, single_cycle=False)
    print(f"greedy test at p={p}: mean lifetime {np.mean(th.history['episode_lifetime']):.1f} (single physical qubit: {1 / p:.0f})")
