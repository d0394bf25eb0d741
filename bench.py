#!/usr/bin/env python3
"""Benchmark of the DeepQ-Decoding hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode loop|env] [--config c3]

One "step" = one pass of the hot path over one batch: every lattice of the batch (4096 per GPU at the
headline config c3: d=5 depolarising p=0.011 with faulty syndromes, depth 5) receives an action, the
batched environment kernel steps them, the transition lands in the device replay ring, and (mode
`loop`) one DQN minibatch update runs.  Rank 0 prints ONE JSON line (contract in the task statement):
`value` = whole-job env steps/s with all inputs resident in HBM; `roofline` is for the dominant kernel;
`cpu_baseline` times the CPU oracle (a port; the reference's Python cannot travel to the GPU box) on
this host's cores for a bounded sample.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c1": dict(d=3, error_model="X", use_Y=False, volume_depth=3, p_phys=0.005, p_meas=0.005, n_envs=1),
    "c2": dict(d=5, error_model="X", use_Y=False, volume_depth=5, p_phys=0.007, p_meas=0.007, n_envs=4096),
    "c3": dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011, n_envs=4096),
    "c5": dict(d=7, error_model="DP", use_Y=False, volume_depth=7, p_phys=0.005, p_meas=0.005, n_envs=1024),
}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32-input MFMA dense peak


def env_bytes_per_step(cfg):
    """B_env = 2*S + 4 + 4 + 1 + 8*ceil(|A|/64) + C*(2d+1)^2 (uint8 observation)."""
    d = cfg["d"]
    layers = 1 if cfg["error_model"] == "X" else (3 if cfg["use_Y"] else 2)
    n_act = layers * d * d + 1
    S = 96 if d <= 5 else 160
    return 2 * S + 4 + 4 + 1 + 8 * ((n_act + 63) // 64) + (cfg["volume_depth"] + layers) * (2 * d + 1) ** 2


def cpu_baseline_env(cfg, seconds=10.0):
    """C oracle (oracle/env_oracle.c, a port of the reference's Environments.py) on ONE host core:
    same lattices, same uniform-over-legal policy, bounded to ~`seconds`."""
    from oracle import c_oracle
    n = min(cfg["n_envs"], 4096)
    kw = {k: v for k, v in cfg.items() if k != "n_envs"}
    env = c_oracle.COracleEnv(n_envs=n, **kw)
    env.reset()
    t0, steps = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        a = env.policy_uniform_legal(steps)
        env.step(a, auto_reset=True)
        steps += 1
    dt = time.perf_counter() - t0
    return dict(value=n * steps / dt, unit="env_steps/s", cores=1, kind="port",
                sample=f"{steps} vector steps x {n} lattices, C oracle env + uniform-legal policy, {dt:.1f}s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--mode", default="auto", choices=["auto", "loop", "env"])
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--minibatch", type=int, default=0, help="DQN minibatch per rank (default: n_envs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # one rank per GPU; DQ_DIST_BACKEND=gloo (+ several ranks on one GPU) exists only to exercise the multi-rank code path on a
    # single-GPU box -- its numbers mean nothing
    backend = os.environ.get("DQ_DIST_BACKEND", "nccl")
    device = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    dq = importlib.import_module("deepq-decoding_amd")
    cfg = dict(CONFIGS[args.config])
    n_local = cfg.pop("n_envs")
    mode = args.mode
    if mode == "auto":
        mode = "loop" if hasattr(dq, "bench_loop") or _has_agent() else "env"

    if mode == "env":
        runner = EnvOnly(dq, cfg, n_local, rank)
    else:
        runner = importlib.import_module("deepq-decoding_amd.bench_loop").FullLoop(
            dq, cfg, n_local, rank, world, args.minibatch or n_local)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        runner.step(timed=False)
    if hasattr(runner, "pick_dominant"):
        runner.pick_dominant()          # untimed probe: which kernel family takes the most time per step
        runner.arm(args.steps)          # ... that family's launches in the timed region are bracketed by HIP events
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step(timed=True)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    replicas_identical = None
    if world > 1 and hasattr(runner, "core"):
        # outside the timed region: every rank must hold bit-identical parameters (same all-reduced gradient, same Adam step)
        chk = runner.core.params.view(torch.int32).to(torch.int64).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(lo.item() == hi.item())
        if not replicas_identical and rank == 0:
            print("WARNING: the ranks' parameters diverged", file=sys.stderr)

    if rank == 0:
        out = {
            "metric": "env steps/sec + DQN updates/sec, d=5 depolarising, batch 4096, 1/2/4/8 GPU",
            "value": n_local * world * args.steps / dt,
            "unit": "env_steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": runner.dtype,
            "data": "synthetic",
            "config": dict(workload=f"{args.config}: d={cfg['d']} {cfg['error_model']} p_phys=p_meas={cfg['p_phys']} depth={cfg['volume_depth']}, "
                                    f"{n_local} lattices/GPU, mode={mode}", **runner.config()),
        }
        out.update(runner.report(args.steps, dt, world))
        if replicas_identical is not None:
            out["replicas_identical"] = replicas_identical
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = runner.cpu_baseline(dict(cfg, n_envs=n_local))
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def _has_agent():
    try:
        importlib.import_module("deepq-decoding_amd.bench_loop")
        return True
    except ImportError:
        return False


class EnvOnly:
    """Environment kernel + device policy only (uniform over legal actions, auto-reset)."""
    dtype = "u64 bit-planes / u8"

    def __init__(self, dq, cfg, n_local, rank):
        import torch
        self.torch = torch
        self.cfg, self.n = cfg, n_local
        self.env = dq.VectorEnv(n_envs=n_local, env_id_base=rank * n_local, **cfg)
        self.env.reset()
        self.action = torch.zeros(n_local, dtype=torch.int32, device="cuda")
        self.t = 0
        self.events = []

    def step(self, timed):
        self.env.select_actions(self.t, out=self.action)
        if timed:
            e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
            e0.record()
        self.env.step(self.action, auto_reset=True)
        if timed:
            e1.record()
            self.events.append((e0, e1))
        self.t += 1

    def config(self):
        return dict(policy="uniform over legal actions (device)", auto_reset=True)

    def report(self, steps, dt, world):
        ms = sum(a.elapsed_time(b) for a, b in self.events) / max(1, len(self.events))
        bytes_per_launch = env_bytes_per_step(self.cfg) * self.n
        achieved = bytes_per_launch / (ms * 1e-3) / 1e9
        return {"roofline": dict(kernel="env_kernel", bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=achieved / HBM_PEAK_GBS, traffic=None, avg_launch_us=ms * 1e3,
                                 algorithmic_bytes_per_launch=bytes_per_launch)}

    def cpu_baseline(self, cfg):
        return cpu_baseline_env(cfg)


if __name__ == "__main__":
    main()
