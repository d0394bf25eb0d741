import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
dq = importlib.import_module("deepq-decoding_amd")
Q = importlib.import_module("deepq-decoding_amd.qnet")
from oracle import dqn_oracle as O
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
B, A, R = 100, 51, 700
shape = (7, 11, 11)
net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, dueling=True, max_batch=B)
params = net.init_params((11, 22))
rng = np.random.RandomState(5)
flat = params.cpu().numpy().copy(); flat += (rng.randn(flat.size) * 0.02).astype(np.float32); params.copy_(torch.from_numpy(flat))
obs = (rng.rand(B, *shape) < 0.3).astype(np.uint8)
cu = lambda a: torch.from_numpy(a).cuda()
obs_t = cu(obs)
q1o, q1t = (cu(rng.randn(B, A).astype(np.float32)) for _ in range(2))
reward, terminal = cu((rng.rand(R) < 0.4).astype(np.float32)), cu((rng.rand(R) < 0.2).astype(np.uint8))
action, idx = cu(rng.randint(0, A, size=R).astype(np.int32)), cu(rng.randint(0, R, size=B).astype(np.int32))
out = {}
for name in ("separate", "one"):
    p_ = params.clone(); g_ = torch.empty_like(params)
    met = torch.full((Q.TD_METRICS_FLOATS,), 7.0, dtype=torch.float32, device="cuda")
    y, dq_ = torch.empty(B, device="cuda"), torch.empty((B, A), device="cuda")
    q0 = net.forward(p_, obs_t, training=True, seed=(1, 2), t=1)
    td = dict(q_online_s1=q1o, q_target_s1=q1t, q_s0=q0, reward=reward, terminal=terminal, action=action, gamma=0.99,
              grad_scale=1.0 / B, index=idx, y=y, dq=dq_, metrics=met)
    if name == "separate":
        Q.td_update(q1o, q1t, q0, reward, terminal, action, 0.99, grad_scale=1.0 / B, index=idx, y=y, dq=dq_, metrics=met)
        net.set_grad_scale(1.0 / B)
        net.backward(p_, dq_, grads=g_)
        net.set_grad_scale(0.0)
    else:
        m_, v_ = torch.zeros_like(params), torch.zeros_like(params)
        net.td_backward_phase0(p_, td, g_)
        net.backward_phase(p_, dq_, g_, 1)
    out[name] = [x.clone().cpu().numpy() for x in (y, dq_, g_)]
spec = O.QNetSpec(shape, C_LAYERS, FF_LAYERS, A, dueling=True)
for k, nm in enumerate(("y", "dq", "g")):
    a, b = out["separate"][k], out["one"][k]
    print(nm, "equal", np.array_equal(a, b), "max diff", np.abs(a - b).max(), "n diff", int((a != b).sum()), "of", a.size)
ga, gb = spec.split(out["separate"][2]), spec.split(out["one"][2])
for li, ((ka, ba), (kb, bb)) in enumerate(zip(ga, gb)):
    print("layer", li, "kernel n diff", int((ka != kb).sum()), "max", np.abs(ka - kb).max(), "| bias n diff", int((ba != bb).sum()))
