"""Where a 20-step timed region loses time against a long one (development aid; GPU box): per-step HIP-event timestamps of the first steps behind a
synchronisation, and the host's own clock around the region."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
bl = importlib.import_module("deepq-decoding_amd.bench_loop")
cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
loop = bl.FullLoop(dq, cfg, 4096, 0, 1, 4096)
for _ in range(5):
    loop.step(timed=False)
for rep in range(3):
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(22)]
    t0 = time.perf_counter()
    ev[0].record()
    for k in range(20):
        loop.step(timed=True)
        ev[k + 1].record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    per = [1e3 * ev[k].elapsed_time(ev[k + 1]) for k in range(20)]
    print(f"rep {rep}: host region {1e6 * (t2 - t0) / 20:.1f} us/step (enqueue done after {1e6 * (t1 - t0):.0f} us, sync returned {1e6 * (t2 - t1):.0f} us later); "
          f"event span {1e3 * ev[0].elapsed_time(ev[20]) / 20:.1f} us/step; per step: " + " ".join(f"{x:.0f}" for x in per))
