"""Training forward + backward of the c3 network at batch 4096, a few times (development aid for rocprofv3 counter passes: tools/pmc_any.sh)."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
shape, A, batch = (7, 11, 11), 51, 4096
net = dq.QNetwork(shape, [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], A, max_batch=batch)
params = net.init_params((11, 22))
rng = np.random.RandomState(5)
obs = torch.from_numpy((rng.rand(batch, *shape) < 0.3).astype(np.uint8)).cuda()
dqt = torch.from_numpy((rng.randn(batch, A) / batch).astype(np.float32)).cuda()
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    net.forward(params, obs, training=True, seed=(1, 2), t=3)
    net.backward(params, dqt)
torch.cuda.synchronize()
