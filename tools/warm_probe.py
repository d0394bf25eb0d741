"""Each launch group of the vector step repeated ON ITS OWN (development aid; GPU box): the same kernels as in the headline loop (c3, 4096 lattices),
but back to back with themselves, i.e. with their code and their weights warm in the instruction caches / L2.  Prints microseconds per repetition;
compare with the in-loop averages of tools/prof_loop.sh: the difference is what a launch pays for starting cold."""
import importlib, os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dq = importlib.import_module("deepq-decoding_amd")
bl = importlib.import_module("deepq-decoding_amd.bench_loop")
from importlib import import_module
_dist = import_module("deepq-decoding_amd.dist")
cfg = dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011)
loop = bl.FullLoop(dq, cfg, 4096, 0, 1, 4096)
for _ in range(100):
    loop.step(timed=False)
core = loop.core
net = core.net
t = core.updates + 1
sb = _dist.shard(0, core.N, core.batch_size)[1]
jobs = core._update_jobs(t, sb)
jobs.append(core._obs_job(params=core.params, slot=core.cur, batch=core.N, out=core.q_act, packed=core.params_pk))


def timed(name, fn, n=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {1e3 * e0.elapsed_time(e1) / n:.2f} us per repetition")


timed("forwards (conv chain + dense chain, 4 jobs)", lambda: net.forward_multi(jobs))
net.forward_multi(jobs)
td = core._td_job()
timed("dense backward + TD, dense weight gradients, dense reduction (phase 0, no riders)", lambda: net.td_backward_phase0(core.params, td, core.grads))
timed("conv backward + conv reduction (phase 1)", lambda: net.backward_phase(core.params, core.dq, core.grads, 1))
timed("pack", lambda: net.pack(core.params, out=core.params_pk) if "out" in net.pack.__code__.co_varnames else net.pack(core.params))
timed("whole step", lambda: loop.step(timed=False))
