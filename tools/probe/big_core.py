import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
dq = importlib.import_module("deepq-decoding_amd")
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
seed = (0x5EED, 0xD0DEC0DE)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
env = dq.VectorEnv(n_envs=N, env_id_base=0, seed=seed, **cfg)
net = dq.QNetwork(env.obs_shape, C_LAYERS, FF_LAYERS, env.num_actions, max_batch=N)
core = dq.DQNCore(env, net, batch_size=N, memory_limit=N * 7, gamma=0.99, lr=1e-3, seed=seed, rank=0, world_size=1)
core.reset_env()
for _ in range(6):
    core.act_and_step(0.3)
torch.cuda.synchronize(); print("acting ok", flush=True)
core._join_env()
_q = importlib.import_module("deepq-decoding_amd.qnet")
t = core.updates + 1
_q.replay_sample(core.terminal_ring, core.N, core.T, core.cur, core.filled, core.batch_size, core.seed, t, sample_base=0, out=core.index)
torch.cuda.synchronize(); print("sample ok", flush=True)
net.forward_multi(core._update_jobs(t, 0))
torch.cuda.synchronize(); print("forward_multi ok", flush=True)
net.td_backward_phase0(core.params, core._td_job(), core.grads)
torch.cuda.synchronize(); print("phase0 ok", flush=True)
net.backward_phase(core.params, core.dq, core.grads, 1)
torch.cuda.synchronize(); print("phase1 ok", float(core.grads.abs().max()), flush=True)
