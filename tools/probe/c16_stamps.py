"""Phase stamps of conv_bwd16_kernel, workgroup 9 (GPU box; tools/probe/build_c16_stamps.sh, then DQ_LIB_PATH=tools/probe/stamps/c16.so python tools/probe/c16_stamps.py)"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
dq = importlib.import_module("deepq-decoding_amd")
E = importlib.import_module("deepq-decoding_amd.env")
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
d, depth, layers, A = 5, 5, 2, 51
net = dq.QNetwork((depth + layers, 2 * d + 1, 2 * d + 1), [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]], A, max_batch=batch)
net.set_patch_input(depth, E.patch_stride_words(d))
params = net.init_params((11, 22))
pk = net.pack(params)
rng = np.random.RandomState(5)
patch = torch.from_numpy(rng.randint(0, 1 << (4 * depth + layers), size=(batch, E.patch_stride_words(d))).astype(np.int32)).cuda()
dqt = torch.from_numpy((rng.randn(batch, A) / batch).astype(np.float32)).cuda()
for _ in range(10):
    net.forward_multi([dict(params=params, obs=patch, patch=True, packed=pk, training=True, seed=(1, 2), t=3)])
    net.backward(params, dqt)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (32 * 16))()
dq.lib().dq_dbg_read_c16(buf)
t = np.array(buf, dtype=np.int64).reshape(32, 16)
names = {0: "entry", 1: "copies issued, tables stored", 26: "behind the group loop", 27: "dW2 / dW3 partials stored", 28: "end"}
ph = ["top", "copies landed + barrier", "patch image, g3 sums", "dW3", "barrier", "g2", "barrier", "g2 sums + dW2", "barrier", "g1", "barrier", "g1 sums + dW1"]
for g in range(2):
    for i, n in enumerate(ph):
        names[2 + 12 * g + i] = "group %d: %s" % (g, n)
order = [k for k in sorted(names) if t[k].max() > 0]
prev = None
print("%-36s %8s %8s %8s   (cycles since the previous stamp: fastest wave, median, slowest; then the slowest wave's cycles since entry)" % ("", "min", "median", "max"))
for k in order:
    if prev is not None:
        dlt = t[k] - t[prev]
        print("%-36s %8d %8d %8d   %8d" % (names[k], dlt.min(), np.median(dlt), dlt.max(), (t[k] - t[0]).max()))
    prev = k
