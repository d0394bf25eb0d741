import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
dq = importlib.import_module("deepq-decoding_amd")
bl = importlib.import_module("deepq-decoding_amd.bench_loop")
cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
loop = bl.FullLoop(dq, cfg, 4096, 0, 1, 4096)
for _ in range(200): loop.step(timed=False)
torch.cuda.synchronize()
for n in (50, 200, 1000):
    t0 = time.perf_counter()
    for _ in range(n): loop.step(timed=False)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: host enqueue {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
