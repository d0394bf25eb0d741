"""Does a hipGraph help the vector step?  (development aid, GPU box)

    python tools/graph_probe.py [steps=2000]

Captures ONE DQNCore.step_and_update (7 launches) into a torch.cuda.CUDAGraph (= hipGraph) and replays it: the replay repeats the same
ring slots and counters, so its RESULTS are meaningless, but it launches exactly the step's kernels on exactly the step's sizes --
what is measured is the launch path.  Printed beside the ordinary loop (host enqueues 7 launches per step through ctypes)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
bl = importlib.import_module("deepq-decoding_amd.bench_loop")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
loop = bl.FullLoop(dq, cfg, 4096, 0, 1, 4096)
for _ in range(50):
    loop.step(False)
torch.cuda.synchronize()


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


us_loop = timed(lambda: loop.core.step_and_update(0.1), steps)
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(side):
    loop.core.step_and_update(0.1)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        loop.core.step_and_update(0.1)
torch.cuda.synchronize()
us_graph = timed(g.replay, steps)
print(f"ordinary loop: {us_loop:.1f} us per vector step; hipGraph replay of one captured step: {us_graph:.1f} us per step")
