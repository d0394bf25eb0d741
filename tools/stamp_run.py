import importlib, os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
dq = importlib.import_module("deepq-decoding_amd")
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
shape, A = (7, 11, 11), 51
for batch in (4096, 8192):
    net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, max_batch=batch)
    params = net.init_params((11, 22))
    obs = torch.from_numpy((np.random.RandomState(5).rand(batch, *shape) < 0.3).astype(np.uint8)).cuda()
    for _ in range(5):
        net.forward(params, obs)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    dq.lib().dq_dbg_read(buf)
    t = [buf[i] for i in range(9)]
    print(batch, "phase cycles: stage, dense1, b2-issue, epilogue, dense2, reduce, dense3, head:", [t[i + 1] - t[i] for i in range(8)], "total", t[8] - t[0])
