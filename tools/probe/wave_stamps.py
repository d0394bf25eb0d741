"""Wall-clock timeline of conv_wave_kernel's waves (GPU box; library built with DQ_EXTRA_FLAGS=-DCW_STAMPS):  python tools/probe/wave_stamps.py [batch]"""
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
kw = dict(patch=True, packed=pk)
four = lambda: net.forward_multi([dict(params=params, obs=patch, **kw), dict(params=params, obs=patch, **kw),
                                  dict(params=params, obs=patch, training=True, seed=(1, 2), t=3, **kw), dict(params=params, obs=patch, **kw)])
for _ in range(20): four()
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (8 * 4096))()
dq.lib().dq_dbg_read_wave(buf)
t = np.array(buf, dtype=np.int64).reshape(8, 256, 16)
live = t[3] > 0
t0 = t[0][live].min()
f = lambda x: (x - t0) / 100.0
print("waves alive:", int(live.sum()))
for i, name in enumerate(("entry", "behind the prologue barrier", "behind the first pair", "end")):
    v = f(t[i][live & (t[i] > 0)])
    print("%-28s first %.2f  median %.2f  p90 %.2f  last %.2f us" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
wg_end = np.where(live, t[3], 0).max(axis=1)
wg_start = np.where(live, t[0], 1 << 62).min(axis=1)
print("workgroup spans: median %.2f  max %.2f us;  ends: first %.2f last %.2f" % (np.median(wg_end - wg_start) / 100.0, (wg_end - wg_start).max() / 100.0, f(wg_end.min()), f(wg_end.max())))
pro = (t[1] - t[0])[live] / 100.0
print("prologue (entry -> barrier): median %.2f max %.2f us;  first pair: median %.2f us;  rest: median %.2f us" %
      (np.median(pro), pro.max(), np.median((t[2] - t[1])[live & (t[2] > 0)]) / 100.0, np.median((t[3] - t[2])[live & (t[2] > 0)]) / 100.0))
for i, name in ((4, "weight copies issued"), (5, "per-lane constants computed"), (6, "own copies landed (vmcnt 0)")):
    v = f(t[i][live & (t[i] > 0)])
    print("%-28s first %.2f  median %.2f  p90 %.2f  last %.2f us" % (name, v.min(), np.median(v), np.percentile(v, 90), v.max()))
cyc = t[7][live & (t[7] > 0)].astype(np.float64); us = ((t[3] - t[0])[live & (t[7] > 0)]) / 100.0
print("shader clock over the waves' lives: median %.0f MHz (cycles / wall-clock us)" % np.median(cyc / us))
