import importlib, os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
dq = importlib.import_module("deepq-decoding_amd")
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
shape, A, batch = (7, 11, 11), 51, 4096
net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, max_batch=batch)
params = net.init_params((11, 22))
rng = np.random.RandomState(5)
obs = torch.from_numpy((rng.rand(batch, *shape) < 0.3).astype(np.uint8)).cuda()
dqt = torch.from_numpy((rng.randn(batch, A) / batch).astype(np.float32)).cuda()
for _ in range(5):
    net.forward(params, obs, training=True, seed=(1, 2), t=3)
    net.backward(params, dqt)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
dq.lib().dq_dbg_read(buf)
for w in range(2):
    t = [buf[i * 2 + w] for i in range(7)]
    print("wave", w, "dense_bwd phases: dueling, gY2, gH1, barrier, gX-loop, gX-epilogue:", [t[i + 1] - t[i] for i in range(6)], "total", t[6] - t[0])
