"""A/B of the forward's convolution kernels on the GPU box (development aid, not part of the product).

    python tools/conv_ab.py [config=c3] [batch=4096] [iters=200]

Times dq_qnet_forward_multi (the step's four forwards: three inference jobs + one training job, and one inference job alone) with the
wave pipeline (csrc/conv_pipe.hip) and with conv_chain_kernel (csrc/fused.hip), per kernel through the library's own HIP-event timers
(dq_prof_*), and prints the largest difference of the two kernels' outputs.  With a -DDQ_STAMPS=5 build (tools/build_stamps.sh 5,
DQ_LIB_PATH=tools/probe/stamps/s5.so) it also prints the pipeline's per-step cycle stamps of workgroup DQ_STAMP_BLOCK."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
SHAPES = {"c1": ((4, 7, 7), 10), "c2": ((6, 11, 11), 26), "c3": ((7, 11, 11), 51), "c5": ((9, 15, 15), 99)}
name = sys.argv[1] if len(sys.argv) > 1 else "c3"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 200
shape, A = SHAPES[name]
net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, max_batch=batch)
params = net.init_params((11, 22))
target = net.init_params((5, 6))
pk, tpk = net.pack(params), net.pack(target)
rng = np.random.RandomState(5)
ring = torch.from_numpy((rng.rand(3 * batch, *shape) < 0.3).astype(np.uint8)).cuda()
idx = torch.from_numpy(rng.randint(0, 2 * batch, size=batch).astype(np.int32)).cuda()
outs = [torch.zeros((batch, A), device="cuda") for _ in range(4)]


def jobs4():
    return [dict(params=target, obs=ring, batch=batch, index=idx, index_off=batch, index_mod=3 * batch, out=outs[0], packed=tpk),
            dict(params=params, obs=ring, batch=batch, index=idx, index_off=batch, index_mod=3 * batch, out=outs[1], packed=pk),
            dict(params=params, obs=ring, batch=batch, index=idx, training=True, seed=(1, 2), t=3, sample_base=0, out=outs[2], packed=pk),
            dict(params=params, obs=ring[:batch], batch=batch, out=outs[3], packed=pk)]


def timed(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


res = {}
for legacy in (False, True):
    net.set_fused(True, conv_pipe=not legacy)
    tag = "conv_chain" if legacy else "conv_pipe "
    t4 = timed(lambda: net.forward_multi(jobs4()), iters)
    t1 = timed(lambda: net.forward_multi(jobs4()[3:]), iters)
    net.forward_multi(jobs4())
    torch.cuda.synchronize()
    dqt = torch.from_numpy((np.random.RandomState(7).randn(batch, A) / batch).astype(np.float32)).cuda()
    res[legacy] = [o.cpu().numpy().copy() for o in outs] + [net.backward(params, dqt).cpu().numpy().copy()]    # (gradient: reads the saved a1 / a2)
    print(f"[{tag}] four jobs (conv + dense launches) {t4:7.1f} us   one inference job {t1:7.1f} us")
for i, (x, y) in enumerate(zip(res[False], res[True])):
    print(f"output {i}: max |pipe - chain| = {np.abs(x - y).max():.3e} (scale {np.abs(y).max():.3e})")

if hasattr(dq.lib(), "dq_dbg_read_pipe"):
    net.set_fused(True, conv_pipe=True)
    for _ in range(3):
        net.forward_multi(jobs4())
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 4096)()
    dq.lib().dq_dbg_read_pipe(buf)
    names = ["setup"] + [f"s{t}:{p}" for t in range(-1, 4) for p in ("a", "b", "bar")]
    for w in range(8):
        t = [buf[i * 8 + w] for i in range(17)]
        print("wave", w, "setup: DMA issued (A), at barrier 1, at barrier 2:", int(buf[18 * 8 + w] - t[0]) if w < 4 else 0, int(buf[17 * 8 + w] - t[0]), int(buf[19 * 8 + w] - t[0]), [int(t[i + 1] - t[i]) for i in range(16)], "total", int(t[16] - t[0]))
    print(names, "(A waves: a = conv1, b = DMA wait + issue; B waves: a = conv2 + conv3; steps >= 3 overwrite slot s3)")
