"""Phase timestamps of one fused kernel (development aid; GPU box).  Build with
    DQ_EXTRA_FLAGS="-DDQ_STAMPS=<tag>" python deepq-decoding_amd/build.py --force      (tags: csrc/common.h DQ_TAG_*)
then    python tools/stamp_run.py <tag>
prints, per wave of workgroup DQ_STAMP_BLOCK, the shader cycles between consecutive DQ_STAMP points."""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
tag = int(sys.argv[1]) if len(sys.argv) > 1 else 4
C_LAYERS, FF_LAYERS = [[64, 3, 2], [32, 2, 1], [32, 2, 1]], [[512, 0.2]]
shape, A, batch = (7, 11, 11), 51, int(os.environ.get("DQ_STAMP_BATCH", 4096))
PD, PDIST, PLAYERS = 5, 5, 2                 # patch words: syndrome planes, distance, action planes
if os.environ.get("DQ_STAMP_CFG") == "c5":   # d = 7 depolarising, depth 7, at its per-GPU size
    shape, A, batch = (9, 15, 15), 99, int(os.environ.get("DQ_STAMP_BATCH", 1024))
    PD, PDIST, PLAYERS = 7, 7, 2
net = dq.QNetwork(shape, C_LAYERS, FF_LAYERS, A, max_batch=batch)
params = net.init_params((11, 22))
rng = np.random.RandomState(5)
obs = torch.from_numpy((rng.rand(batch, *shape) < 0.3).astype(np.uint8)).cuda()
dqt = torch.from_numpy((rng.randn(batch, A) / batch).astype(np.float32)).cuda()
PATCH = bool(os.environ.get("DQ_STAMP_PATCH"))        # the observations as patch words (include/deepq_hip.h dq_env_patch_output)
jkw = {}
if PATCH:
    E = importlib.import_module("deepq-decoding_amd.env")
    stride = E.patch_stride_words(PDIST) if hasattr(E, "patch_stride_words") else 32
    net.set_patch_input(PD, stride)
    obs = E.obs_to_patch(E.patch_to_obs(torch.from_numpy(rng.randint(0, 1 << (4 * PD + PLAYERS), size=(batch, stride)).astype(np.int32)), PDIST, PD, PLAYERS), PDIST, PD, PLAYERS).cuda().contiguous()
    jkw = dict(patch=True)
    _fwd = net.forward
    net.forward = lambda p, o, **kw: net.forward_multi([dict(params=p, obs=o, **jkw, **kw)])[0]
for _ in range(5):
    net.forward(params, obs, training=True, seed=(1, 2), t=3)
    net.backward(params, dqt)
if os.environ.get("DQ_STAMP_INFER"):                  # the LAST forward launch (whose stamps are read) is an inference forward
    net.forward(params, obs)
if os.environ.get("DQ_STAMP_LOOP"):                   # the LAST forward launch pair has the vector step's shape: 3 inference jobs + the training job
    pk = net.pack(params)
    for _ in range(3):
        net.forward_multi([dict(params=params, obs=obs, packed=pk, **jkw), dict(params=params, obs=obs, packed=pk, **jkw),
                           dict(params=params, obs=obs, training=True, seed=(1, 2), t=3, packed=pk, **jkw), dict(params=params, obs=obs, packed=pk, **jkw)])
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
getattr(dq.lib(), 'dq_dbg_read_fwd' if tag % 10 in (1, 2) and tag != 21 else 'dq_dbg_read_bwd')(buf)
if tag == 21:
    dq.lib().dq_dbg_read_bwd(buf)
    n = [256, 392, 392, 256]
    q = [np.array([buf[k * 1024 + i] for i in range(n[k])], dtype=np.int64) for k in range(4)]
    t0 = q[0].min()
    f = lambda x: "%.2f" % ((x - t0) / 100)
    print("dense data gradients: ends first", f(q[0].min()), "median", f(np.median(q[0])), "last", f(q[0].max()), "us")
    print("dense weight gradients: starts first", f(q[1].min()), "last", f(q[1].max()), "; ends first", f(q[2].min()), "median", f(np.median(q[2])), "last", f(q[2].max()), "us")
    print("convolutional backward: starts first", f(q[3].min()), "last", f(q[3].max()), "us")
    tiles = 49
    dur = (q[2] - q[1]) / 100.0
    by_tile = [dur[[b for b in range(392) if (b >> 3) % tiles == t]] for t in range(tiles)]
    print("weight-gradient workgroup durations by tile (us, mean over slices):", " ".join("%d:%.1f" % (t, by_tile[t].mean()) for t in range(tiles)))
    by_slice = [dur[[b for b in range(392) if (b & 7) == x]] for x in range(8)]
    print("by slice:", " ".join("%.1f" % v.mean() for v in by_slice))
elif tag == 20:
    for _ in range(3): net.forward(params, obs)
    torch.cuda.synchronize()
    dq.lib().dq_dbg_read_fwd(buf)
    q = [np.array([buf[k * 1024 + i] for i in range(512 if k < 2 else 256)], dtype=np.int64) for k in range(4)]
    t0 = q[0].min()
    print("conv: first start 0, last start %.2f, first end %.2f, last end %.2f us" % ((q[0].max() - t0) / 100, (q[1].min() - t0) / 100, (q[1].max() - t0) / 100))
    print("dense: first start %.2f, last start %.2f, first end %.2f, last end %.2f us" % tuple((x - t0) / 100 for x in (q[2].min(), q[2].max(), q[3].min(), q[3].max())))
elif tag > 10:
    # wall-clock start / end (10 ns ticks) of workgroups 0..255 of the LAST launch of the kernel
    def report(what):
        getattr(dq.lib(), 'dq_dbg_read_fwd' if tag % 10 in (1, 2) else 'dq_dbg_read_bwd')(buf)
        st = np.array([buf[i] for i in range(256)], dtype=np.int64); en = np.array([buf[256 + i] for i in range(256)], dtype=np.int64)
        t0 = st.min()
        print(what, "start spread %.2f us; end min %.2f median %.2f max %.2f us; per-workgroup duration median %.2f us" %
              ((st - t0).max() / 100, (en - t0).min() / 100, np.median(en - t0) / 100, (en - t0).max() / 100, np.median(en - st) / 100))
    def timed(fn, n=20):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    if tag % 10 in (1, 2):
        print("inference forward (conv + dense launches): %.1f us per call" % timed(lambda: net.forward(params, obs))); report("inference:")
        print("training forward: %.1f us per call" % timed(lambda: net.forward(params, obs, training=True, seed=(1, 2), t=3))); report("training:")
    else:
        report("backward:")
elif tag == 4:
    names = ["stage", "patch-image", "dW3", "bar", "g2(+next w)", "bar", "dW2", "bar", "g1", "bar", "dW1(to next/end)"]
    for w in range(8):
        t = [buf[i * 8 + w] for i in range(11)] + [buf[12 * 8 + w]]
        t2 = [buf[(12 + i) * 8 + w] for i in range(11)] + [buf[24 * 8 + w]]
        print("wave", w, "group 1:", [t[i + 1] - t[i] for i in range(11)], " group 2:", [t2[i + 1] - t2[i] for i in range(11)],
              " wait part of stage:", buf[11 * 8 + w] - t[0], buf[23 * 8 + w] - t2[0])
    print(names)
    for w in range(8):
        print("wave", w, "kernel start -> first group", buf[0 * 8 + w] - buf[25 * 8 + w], "; last group's end -> kernel end", buf[26 * 8 + w] - buf[24 * 8 + w],
              "; whole kernel", buf[26 * 8 + w] - buf[25 * 8 + w], "cycles; prologue: tables", buf[27 * 8 + w] - buf[25 * 8 + w], "constants", buf[28 * 8 + w] - buf[27 * 8 + w],
              "first copies issued", buf[29 * 8 + w] - buf[28 * 8 + w], "to the loop", buf[0 * 8 + w] - buf[29 * 8 + w])
elif tag == 5:
    for w in range(4):
        t = [buf[i * 8 + w] for i in range(27)]
        print("wave", w, "per iteration [wait+split+store, issue+barrier, mfma]:", [[t[1 + 3 * i + k + 1] - t[1 + 3 * i + k] for k in range(3)] for i in range(8)],
              "first", t[1] - t[0], "epilogue", t[26] - t[25], "total", t[26] - t[0])
elif tag == 1 and os.environ.get("DQ_STAMP_PERSIST"):
    # persistent conv chain: 8 stamps per group of the workgroup (1 top barrier passed, 2 table barrier, 3 conv1 done, 4 barrier, 5 conv2 done, 6 barrier, 7 group done)
    for w in range(4):
        for g in range(4):
            t = [buf[(8 * g + i) * 8 + w] for i in range(1, 8)]
            prev = buf[(8 * (g - 1) + 7) * 8 + w] if g else buf[0 * 8 + w]
            print("wave", w, "group", g, "top wait+barrier", t[0] - prev, "| dma issue + table + bar, conv1, bar, conv2, bar, conv3 (+ w1 + burst):", [t[i + 1] - t[i] for i in range(6)], "group total", t[6] - prev)
elif tag == 1:
    for w in range(4):
        t = [buf[i * 8 + w] for i in range(8)]
        print("wave", w, "w1-issue, stage, conv1, bar, conv2, bar, conv3:", [t[i + 1] - t[i] for i in range(7)], "total", t[7] - t[0])
elif tag == 2:
    for w in range(8):
        t = [buf[i * 8 + w] for i in range(9)]
        print("wave", w, "stage, dense1, w2-issue, pass 0 (epilogue + Dense(|A|) partials), its reduction, pass 1, dueling layer + bar, head (the stamps themselves add waits: each is a global store):", [t[i + 1] - t[i] for i in range(8)], "total", t[8] - t[0])
elif tag == 3:
    for w in range(8):
        t = [buf[i * 8 + w] for i in range(6)]
        print("wave", w, "zero+dueling, gY2, gH1, bar, gX:", [t[i + 1] - t[i] for i in range(5)], "total", t[5] - t[0])
