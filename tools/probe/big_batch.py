import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
dq = importlib.import_module("deepq-decoding_amd")
shape, A = (7, 11, 11), 51
for batch in (4096, 8192, 16384, 32768):
    net = dq.QNetwork(shape, [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], A, max_batch=batch)
    params = net.init_params((11, 22))
    rng = np.random.RandomState(5)
    obs = torch.from_numpy((rng.rand(batch, *shape) < 0.3).astype(np.uint8)).cuda()
    dqt = torch.from_numpy((rng.randn(batch, A) / batch).astype(np.float32)).cuda()
    q = net.forward(params, obs, training=True, seed=(1, 2), t=3)
    torch.cuda.synchronize(); print(batch, "forward ok", float(q.abs().max()), flush=True)
    for phase in (0, 1):
        g = torch.zeros(net.n_params, device="cuda")
        net.backward_phase(params, dqt, g, phase)
        torch.cuda.synchronize(); print(batch, "backward phase", phase, "ok", float(g.abs().max()), flush=True)
