"""Wall-clock start / end of the dense forward's 256 workgroups in the vector step's launch shape (3 inference jobs + the training job):
    DQ_LIB_PATH=tools/probe/stamps/s12.so python tools/probe/dense_fwd_timeline.py"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
dq = importlib.import_module("deepq-decoding_amd")
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
shape, A, batch = (7, 11, 11), 51, 4096
net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, max_batch=batch)
params = net.init_params((11, 22))
rng = np.random.RandomState(5)
obs = torch.from_numpy((rng.rand(batch, *shape) < 0.3).astype(np.uint8)).cuda()
pk = net.pack(params)
for _ in range(5):
    net.forward_multi([dict(params=params, obs=obs, packed=pk), dict(params=params, obs=obs, packed=pk),
                       dict(params=params, obs=obs, training=True, seed=(1, 2), t=3, packed=pk), dict(params=params, obs=obs, packed=pk)])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
dq.lib().dq_dbg_read_fwd(buf)
st = np.array([buf[i] for i in range(256)], dtype=np.int64); en = np.array([buf[256 + i] for i in range(256)], dtype=np.int64)
t0 = st.min()
for name, sl in (("job 0 (inference)", slice(0, 64)), ("job 1 (inference)", slice(64, 128)), ("job 2 (TRAINING)", slice(128, 192)), ("job 3 (inference)", slice(192, 256))):
    s_, e_ = (st[sl] - t0) / 100.0, (en[sl] - t0) / 100.0
    print(f"{name}: start median {np.median(s_):.2f} max {s_.max():.2f}; end min {e_.min():.2f} median {np.median(e_):.2f} max {e_.max():.2f} us; duration median {np.median(e_ - s_):.2f}")
print("launch span %.2f us" % ((en.max() - t0) / 100.0))
