"""One-box A/B of the two observation forms (padded uint8 images / patch words) on the Q-network kernels alone (GPU box):
    python tools/conv_ab.py [batch]          event-timed: the step's four forwards (conv + dense launches), training forward + backward
Under rocprofv3 --kernel-trace --stats the per-kernel averages separate conv_chain_pkernel<4> / <0> and conv_bwd_chain_kernel<4, false> / <2, true>."""
import importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
E = importlib.import_module("deepq-decoding_amd.env")
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
d, depth, layers, A = (5, 5, 2, 51) if os.environ.get("DQ_AB_CFG", "c3") == "c3" else (7, 7, 2, 99)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
shape = (depth + layers, 2 * d + 1, 2 * d + 1)
net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, max_batch=batch)
net.set_patch_input(depth, E.patch_stride_words(d))
params = net.init_params((11, 22))
pk = net.pack(params)
rng = np.random.RandomState(5)
words = torch.from_numpy(rng.randint(0, 1 << (4 * depth + layers), size=(batch, E.patch_stride_words(d))).astype(np.int32))
obs = E.patch_to_obs(words, d, depth, layers)
patch = E.obs_to_patch(obs, d, depth, layers).cuda().contiguous()
obs = obs.cuda().contiguous()
dqt = torch.from_numpy((rng.randn(batch, A) / batch).astype(np.float32)).cuda()


def timed(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for rep in range(2):
    for name, o, kw in (("uint8", obs, {}), ("patch", patch, dict(patch=True))):
        four = lambda: net.forward_multi([dict(params=params, obs=o, packed=pk, **kw), dict(params=params, obs=o, packed=pk, **kw),
                                          dict(params=params, obs=o, training=True, seed=(1, 2), t=3, packed=pk, **kw), dict(params=params, obs=o, packed=pk, **kw)])
        t4 = timed(four)
        g = torch.empty(net.n_params, dtype=torch.float32, device="cuda")
        four()
        tb = timed(lambda: net.backward(params, dqt, g))
        print("%s: four forwards (conv + dense) %.1f us; backward (all launches) %.1f us" % (name, t4, tb))
