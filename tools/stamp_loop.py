"""Stamps of the dense backward's launch INSIDE the vector step (development aid; GPU box): the headline loop (bench_loop.FullLoop, c3) runs
a few steps on a library built with -DDQ_STAMPS=3 (phase cycles of workgroup DQ_STAMP_BLOCK, TD prologue and riding environment step
included), -DDQ_STAMPS=6 -DDQ_STAMP_BLOCK=<a rider block, e.g. 300> (phases of a riding environment workgroup, env_dev.h env_block2) or -DDQ_STAMPS=23 (wall-clock start / end of every workgroup of the launch, riders included).
    DQ_LIB_PATH=tools/probe/stamps/s23.so python tools/stamp_loop.py 23"""
import ctypes, importlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
bl = importlib.import_module("deepq-decoding_amd.bench_loop")
tag = int(sys.argv[1]) if len(sys.argv) > 1 else 23
cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
n = 4096
if os.environ.get("DQ_STAMP_CFG") == "c5":          # the d = 7 configuration at its per-GPU size
    cfg, n = dict(d=7, error_model="DP", use_Y=False, volume_depth=7, p_phys=0.005, p_meas=0.005), 1024
loop = bl.FullLoop(dq, cfg, n, 0, 1, n)
for _ in range(60):
    loop.step(timed=False)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 4096)()
dq.lib().dq_dbg_read_bwd(buf)
if tag == 6:
    names = ["loads at the top (tables, record, Q row, referee tables -> LDS)", "policy", "step (referee look-ups)", "noise rounds", "record + scalar outputs",
             "observation planes -> LDS stage", "barrier", "bookkeeping atomics + stage -> global", "replay sampling"]
    for w in range(8):
        t = [buf[i * 8 + w] for i in range(10)]
        print("wave", w, [t[i + 1] - t[i] for i in range(9)], "total", t[9] - t[0])
    print(names)
elif tag == 3:
    for w in range(8):
        t = [buf[i * 8 + w] for i in range(6)]
        print("wave", w, "zero+TD+dueling, gY2, gH1, bar, gX:", [t[i + 1] - t[i] for i in range(5)], "total", t[5] - t[0])
        u = [buf[i * 8 + w] for i in (0, 6, 7, 8, 9, 1)]
        print("        prologue: preloads issued + LDS zero + barrier, TD stage-1 loads issued, arg-max + stage 2 issued, y / loss, dueling + splits + stores, barrier + metrics:",
              [u[i + 1] - u[i] for i in range(5)])
else:
    st = np.array([buf[i] for i in range(1024)], dtype=np.int64); en = np.array([buf[1024 + i] for i in range(1024)], dtype=np.int64)
    n = int((st > 0).sum())
    t0 = st[:n].min()
    f = lambda x: (x - t0) / 100.0
    dense = slice(0, 256)
    rest = slice(256, n)
    print("blocks stamped:", n)
    for name, sl in (("dense workgroups", dense), ("rider workgroups", rest)):
        s_, e_ = f(st[sl]), f(en[sl])
        print(f"{name}: start min {s_.min():.2f} median {np.median(s_):.2f} max {s_.max():.2f} us; end min {e_.min():.2f} median {np.median(e_):.2f} max {e_.max():.2f} us; "
              f"duration median {np.median(e_ - s_):.2f} max {(e_ - s_).max():.2f} us")
    print("launch span %.2f us" % f(en[:n].max()))
