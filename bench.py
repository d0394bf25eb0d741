#!/usr/bin/env python3
"""Benchmark of the DeepQ-Decoding hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode loop|act|learn|env] [--config c3] [--minibatch B]

One "step" = one pass of the hot path over one batch: every lattice of the batch (4096 per GPU at the
headline config c3: d=5 depolarising p=0.011 with faulty syndromes, depth 5) receives an action, the
batched environment kernel steps them, the transition lands in the device replay ring, and (mode
`loop`) one DQN minibatch update runs.  Rank 0 prints ONE JSON line (contract in the task statement):
`value` = whole-job env steps/s with all inputs resident in HBM; `roofline` is for the dominant kernel (its duration: HIP
events bound to a sample of its launches inside the timed region -- the dispatch packets' own timestamps, csrc/prof.hip);
`cpu_baseline` times the CPU oracle (a port; the reference's Python cannot travel to the GPU box) on
this host's cores for a bounded sample.

Modes (SURVEY.md 8d "reported numbers"): `loop` (iv) full loop, the default and the headline; `env` (i) environment kernel
+ uniform-legal device policy; `act` (ii) Q forward + epsilon-greedy + environment step, no learning; `learn` (iii) the
learner alone -- replay sampling, three forwards, TD step, backward, Adam on a pre-filled ring (value = trained samples/s).

`--gpus N` with N > 1 and no torchrun environment re-launches itself as
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` (one rank per GPU over RCCL);
a run whose world size differs from --gpus fails instead of printing a line.
"""
import argparse
import ctypes
import importlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    "c1": dict(d=3, error_model="X", use_Y=False, volume_depth=3, p_phys=0.005, p_meas=0.005, n_envs=1),
    "c2": dict(d=5, error_model="X", use_Y=False, volume_depth=5, p_phys=0.007, p_meas=0.007, n_envs=4096),
    "c3": dict(d=5, error_model="DP", use_Y=False, volume_depth=5, p_phys=0.011, p_meas=0.011, n_envs=4096),
    "c5": dict(d=7, error_model="DP", use_Y=False, volume_depth=7, p_phys=0.005, p_meas=0.005, n_envs=1024),
    # beyond BASELINE.json's configs: 81-qubit lattices on the wide environment with the matching referee inside the step;
    # the Q-network runs on the per-layer kernels (the fused chains cover |A| <= 127); use with --no-cpu-baseline
    "d9": dict(d=9, error_model="DP", use_Y=False, volume_depth=9, p_phys=0.003, p_meas=0.003, n_envs=1024),
}
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32-input MFMA dense peak
METRIC = "env steps/sec + DQN updates/sec, d=5 depolarising, batch 4096, 1/2/4/8 GPU"


def env_bytes_per_step(cfg):
    """B_env = 2*S + 4 + 4 + 1 + 8*ceil(|A|/64) + C*(2d+1)^2 (uint8 observation)."""
    d = cfg["d"]
    layers = 1 if cfg["error_model"] == "X" else (3 if cfg["use_Y"] else 2)
    n_act = layers * d * d + 1
    S = 96 if d <= 5 else 160
    return 2 * S + 4 + 4 + 1 + 8 * ((n_act + 63) // 64) + (cfg["volume_depth"] + layers) * (2 * d + 1) ** 2


# ---- CPU baselines (the only place the oracle is used here: as the thing TIMED beside the GPU, never as the product) ---------------
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


def cpu_env_worker(config, t_start, seconds):
    """One process of the all-cores environment figure: the C oracle environment of `config` (4096 lattices, uniform-legal policy)
    stepped from wall-clock time t_start for `seconds`; prints the lattice-steps done.  No torch import."""
    from oracle import c_oracle
    cfg = {k: v for k, v in CONFIGS[config].items() if k != "n_envs"}
    env = c_oracle.COracleEnv(n_envs=4096, env_id_base=4096 * (os.getpid() & 0xffff), **cfg)
    env.reset()
    while time.time() < t_start:
        time.sleep(0.01)
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        env.step(env.policy_uniform_legal(steps), auto_reset=True)
        steps += 1
    print(json.dumps(dict(lattice_steps=4096 * steps, seconds=time.perf_counter() - t0)), flush=True)


def cpu_baseline_env_all_cores(config, seconds=5.0):
    """the C oracle environment on ALL host cores -- one process per core this process may run on, independent
    lattices, started together, lattice-steps summed.  Returns (env steps/s, cores)."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    t_start = time.time() + 2.0 + 0.01 * cores
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-env-worker", config, repr(t_start), repr(seconds)]
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env) for _ in range(cores)]
    total, longest = 0, 0.0
    for p in procs:
        out, _ = p.communicate(timeout=120 + seconds)
        r = json.loads(out.decode().strip().splitlines()[-1])
        total, longest = total + r["lattice_steps"], max(longest, r["seconds"])
    return total / longest, cores


def cpu_quota_cores():
    """CPU quota of this container's cgroup in cores (None = unlimited / unknown): the affinity mask may list every host core while
    the quota allows far fewer to run at once, which is what bounds the all-cores figure."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        return None


def _cpu_learner(cfg, c_layers, ff_layers):
    import numpy as np
    from oracle import c_oracle, dqn_oracle as O, torch_dqn
    kw = {k: v for k, v in cfg.items() if k != "n_envs"}
    env = c_oracle.COracleEnv(n_envs=cfg["n_envs"], **kw)
    spec = O.QNetSpec(env.obs_shape, c_layers, ff_layers, env.num_actions)
    return env, spec, torch_dqn.TorchDQN(spec, O.glorot_init(spec, (1, 2)).astype(np.float32), lr=1e-4), np


def cpu_baseline_loop(cfg, B, eps, c_layers, ff_layers, mode="loop", seconds=12.0, seconds_b32=6.0):
    """The SAME loop on this host's cores, assembled from the oracles: C-oracle environment (a port of Environments.py, one core)
    + the torch-CPU fp32 restatement of the keras-rl / Keras update (oracle/torch_dqn.py; oneDNN convolutions + autograd on all
    cores -- the closest thing to the reference's TensorFlow-CPU learner that runs here).  Bounded: whole vector steps until
    ~`seconds` have elapsed (at least 2).  A second leg times the learner alone at minibatch 32, the reference's own setting, to
    set beside its recorded 38-42 updates/s."""
    import torch
    env, spec, learner, np = _cpu_learner(cfg, c_layers, ff_layers)
    n, T = cfg["n_envs"], 6
    ring_obs = np.zeros((T, n) + env.obs_shape, np.uint8)
    ring_a, ring_r, ring_t = np.zeros((T, n), np.int64), np.zeros((T, n), np.float32), np.zeros((T, n), np.uint8)
    rng = np.random.RandomState(0)
    ring_obs[0] = env.reset()
    units, rate = ff_layers[0]
    cur, filled, steps, updates, t0 = 0, 1, 0, 0, time.perf_counter()

    def one_update(batch):
        back = rng.randint(2, max(3, filled), size=batch)                  # keras-rl's range: never the newest transition
        e = rng.randint(0, n, size=batch)
        s0 = (cur - back) % T
        s1 = (s0 + 1) % T
        keep = rng.rand(batch, units) >= rate
        learner.update(ring_obs[s0, e], ring_a[s0, e], ring_r[s0, e], ring_t[s0, e], ring_obs[s1, e], keep)

    while steps < 2 or time.perf_counter() - t0 < seconds:
        if mode != "learn" or filled < T:
            with torch.no_grad():
                q = learner.forward(learner.params, ring_obs[cur]).numpy()
            a = np.where(rng.rand(n) < eps, env.policy_uniform_legal(steps), q.argmax(axis=1)).astype(np.int32)
            nxt = (cur + 1) % T
            obs, r, done = env.step(a, auto_reset=True)
            ring_obs[nxt], ring_a[cur], ring_r[cur], ring_t[cur] = obs, a, r, done
            cur, filled = nxt, min(T, filled + 1)
            if mode == "learn" and filled == T:
                steps, t0 = 0, time.perf_counter()                         # ring filled: the learner-only clock starts here
                continue
        if mode != "act" and filled >= 4:
            one_update(B)
            updates += 1
        steps += 1
    dt = time.perf_counter() - t0
    out = dict(value=(B if mode == "learn" else n) * steps / dt, unit="dqn_samples/s" if mode == "learn" else "env_steps/s",
               cores=os.cpu_count(), kind="port", torch_threads=torch.get_num_threads(),
               sample=f"{steps} vector steps x {n} lattices, mode={mode} (C-oracle env step on one core; Q forward"
                      f"{'' if mode == 'act' else f' + one {B}-sample double-DQN update'} in torch-CPU fp32 on all cores), {dt:.1f}s",
               dqn_updates_per_s=(updates / dt) if mode != "act" else None)
    if mode != "act" and seconds_b32 > 0 and filled >= 4:
        threads_all = torch.get_num_threads()
        torch.set_num_threads(min(4, threads_all))                         # the reference's SLURM jobs had 4 cores (GEN:87)
        for _ in range(3):
            one_update(32)
        t1, k = time.perf_counter(), 0
        while k < 3 or time.perf_counter() - t1 < seconds_b32:
            one_update(32)
            k += 1
        out["minibatch32_updates_per_s"] = k / (time.perf_counter() - t1)
        out["minibatch32_threads"] = torch.get_num_threads()
        torch.set_num_threads(threads_all)
        out["reference_recorded_updates_per_s"] = "38.4-41.6 (4 cores, TF-CPU, 2018; trained_models/d5_dp/*/training_history.json)"
    return out


# ---- launch plumbing -----------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n):
    """--gpus N > 1 without a torchrun environment: re-run this script under torch.distributed.run, one rank per GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), OMP_NUM_THREADS="4")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    if len(sys.argv) >= 5 and sys.argv[1] == "--cpu-env-worker":
        return cpu_env_worker(sys.argv[2], float(sys.argv[3]), float(sys.argv[4]))
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--mode", default="loop", choices=["loop", "act", "learn", "env"])
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--minibatch", type=int, default=0, help="DQN minibatch per rank (default: n_envs)")
    ap.add_argument("--lattices", type=int, default=0, help="lattices per rank (default: the configuration's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--env-steps-per-launch", type=int, default=64, help="--mode env: agent steps per environment launch (dq_env_act_steps); 1 = one launch per step")
    ap.add_argument("--env-obs", default="patch", choices=["patch", "uint8"], help="--mode env: successor observations as the loop's patch words or as the reference's uint8 planes")
    ap.add_argument("--dist-probe-steps", type=int, default=50,
                    help="--gpus N > 1: untimed vector steps per form of the gradient exchange (single / overlap) timed before the timed region, which then runs "
                         "in the faster; 0 or an explicit DQ_DIST_MODE: no probing")
    ap.add_argument("--ratio-steps", type=int, default=100,
                    help="vector steps of the second timed leg at the reference's replay ratio (32 trained samples per environment step), run behind the "
                         "headline region of a default 1-GPU loop run and printed as `reference_replay_ratio`; 0 = skip it")
    ap.add_argument("--updates-per-step", type=int, default=1,
                    help="minibatch updates per vector step (loop mode).  The reference trains 32 samples per environment step "
                         "(one 32-sample update per step, Single_Point_Training_Script.py:119-127): --updates-per-step 32 at the default "
                         "minibatch = lattices, or --minibatch 32 --updates-per-step <lattices>")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a run of a different size")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs (no CPU fallback)"
    # one rank per GPU; DQ_DIST_BACKEND=gloo (+ several ranks on one GPU) exists only to exercise the multi-rank code path on a
    # single-GPU box -- its numbers mean nothing
    backend = os.environ.get("DQ_DIST_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        sys.exit(f"bench.py: {world} ranks over RCCL need {world} GPUs, this node has {torch.cuda.device_count()}")
    device = local_rank % torch.cuda.device_count() if backend != "nccl" else local_rank
    torch.cuda.set_device(device)
    # DQ_DIST_FORCE=1 at --gpus 1: a one-rank RCCL group, and the learner takes its several-GPU branch through it (communicator,
    # asynchronous all-reduce behind the convolutional backward, separate Adam) -- the only way to run that code through the real
    # backend on a one-GPU box; the line then carries the distributed fields with rccl_ranks = 1
    force_dist = world == 1 and os.environ.get("DQ_DIST_FORCE") == "1"
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.update(RANK="0", WORLD_SIZE="1")
    json_out = sys.stdout
    if world > 1 or force_dist:
        # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner through C stdio on file descriptor 1, buffered when that
        # is a pipe and flushed at process exit -- i.e. BEHIND rank 0's line, once per rank.  Descriptor 1 is pointed at stderr for everything but
        # the line itself (and the C buffers are flushed on every rank in front of it, below)
        sys.stdout.flush()
        json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == args.gpus

    dq = importlib.import_module("deepq-decoding_amd")
    cfg = dict(CONFIGS[args.config])
    n_local = args.lattices or cfg.pop("n_envs")
    cfg.pop("n_envs", None)
    mode = args.mode

    if mode == "env":
        runner = EnvOnly(dq, cfg, n_local, rank, steps_per_launch=args.env_steps_per_launch, obs_form=args.env_obs, config_name=args.config)
    else:
        runner = importlib.import_module("deepq-decoding_amd.bench_loop").FullLoop(
            dq, cfg, n_local, rank, world, args.minibatch or n_local, mode=mode, config_name=args.config, updates_per_step=max(1, args.updates_per_step))
        runner.pmc_minibatch, runner.pmc_lattices = int(args.minibatch or 0), int(args.lattices or 0)

    def sync():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    if (world > 1 or force_dist) and hasattr(runner, "core") and os.environ.get("DQ_DIST_MODE", "single") in ("single", "overlap"):
        runner.core.ensure_comm()       # the learner's own communicator: its rendezvous belongs to the set-up, also with --warmup 0
    rccl_ranks = 0
    if world > 1 or force_dist:
        # pre-flight (untimed): one all-reduce of a rank-stamped vector, checked against what `world` ranks must give, so that a mis-joined communicator
        # fails loudly here instead of producing a plausible line; rccl_ranks is ncclCommCount's answer, not WORLD_SIZE
        comm = getattr(getattr(runner, "core", None), "_rccl", None)
        if comm is not None:
            rccl_ranks = comm.preflight()
        else:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
            v = (float(rank) + torch.arange(64, dtype=torch.float32, device=dev) / 8.0)
            dist.all_reduce(v)
            want = world * (world - 1) / 2.0 + world * torch.arange(64, dtype=torch.float32, device=dev) / 8.0
            if not torch.equal(v, want):
                raise RuntimeError(f"pre-flight all-reduce over {world} ranks returned {v[:4].tolist()} ..., expected {want[:4].tolist()} ... (rank {rank})")
            rccl_ranks = dist.get_world_size() if dist.get_backend() == "nccl" else 0
    for _ in range(args.warmup):
        runner.step(timed=False)
    dist_tune = None
    if (world > 1 or force_dist) and mode == "loop" and hasattr(runner, "core") and "DQ_DIST_MODE" not in os.environ and args.dist_probe_steps > 0:
        # Self-tuning exchange: which form of the gradient all-reduce wins depends on wire time no 1-GPU box can show (DESIGN.md section 7:
        # `single` up to 4 ranks, about even at 8).  Both are timed here, untimed by the contract -- `dist_probe_steps` vector steps each behind a barrier, MAX over
        # the ranks so that every rank sees the same two numbers -- and the timed region runs in the faster.  An explicit DQ_DIST_MODE switches this off.
        dist_tune = {"probe_steps": args.dist_probe_steps, "us_per_step": {}}
        for m in ("single", "overlap"):
            os.environ["DQ_DIST_MODE"] = m
            runner.core.ensure_comm(with_overlap=(m == "overlap"))
            for _ in range(5):
                runner.step(timed=False)
            sync()
            t0 = time.perf_counter()
            for _ in range(args.dist_probe_steps):
                runner.step(timed=False)
            sync()
            tm = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            dist_tune["us_per_step"][m] = 1e6 * float(tm.item()) / args.dist_probe_steps
        dist_tune["chosen"] = min(dist_tune["us_per_step"], key=dist_tune["us_per_step"].get)
        os.environ["DQ_DIST_MODE"] = dist_tune["chosen"]
        for _ in range(5):
            runner.step(timed=False)
    if hasattr(runner, "pick_dominant"):
        runner.pick_dominant()          # untimed probe: which kernel family takes the most time per step
    if os.environ.get("DQ_BENCH_NO_ARM") != "1":     # (diagnostic: the timed region without any event, no roofline object)
        runner.arm(args.steps)          # ... launches of that family inside the timed region carry a HIP event pair (prof.hip; a sample of them, see prof_stride)
    if (world > 1 or force_dist) and hasattr(runner, "core"):
        runner.core.ar_events = []      # HIP events around the exposed part of the gradient all-reduce, inside the timed loop
        runner.core.ar_stride = importlib.import_module("deepq-decoding_amd.bench_loop").prof_stride(args.steps)      # (a sample of the steps, as the kernel timing)
        runner.core.ar_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps // runner.core.ar_stride + 8)]
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        runner.step(timed=True)
    sync()
    dt = time.perf_counter() - t0
    if world > 1 or force_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if os.environ.get("DQ_BENCH_FAMILIES") == "1" and hasattr(runner, "family_times") and rank == 0:
        extra = runner.report(args.steps, dt, world)       # (collects the timed region's events first)
        fam = runner.family_times()
        print("per-family launch durations in the free-running loop (us): " +
              ", ".join(f"{k} {v['avg_us']:.2f} x{v['per_step']}" for k, v in fam.items()) +
              f"; sum per step {sum(v['avg_us'] * v['per_step'] for v in fam.values()):.1f}", file=sys.stderr)
        runner.report = lambda *a, _e=extra: _e

    replicas_identical, allreduce = None, None
    if (world > 1 or force_dist) and hasattr(runner, "core"):
        # outside the timed region: every rank must hold bit-identical parameters (same all-reduced gradient, same Adam step)
        chk = runner.core.params.view(torch.int32).to(torch.int64).sum().reshape(1)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        replicas_identical = bool(lo.item() == hi.item())
        if not replicas_identical and rank == 0:
            print("WARNING: the ranks' parameters diverged", file=sys.stderr)
        allreduce = allreduce_probe(torch, dist, runner.core, backend)
        ev = runner.core.ar_events or []
        # inside the timed loop, per step: conv-range all-reduce + wait for the dense range's asynchronous one (core.py _learn), i.e.
        # the time the step's critical path spends on communication (what N > 1 adds to the N = 1 step besides the separate Adam launch)
        allreduce["exposed_us_per_step"] = (1e3 * sum(a.elapsed_time(b) for a, b in ev) / len(ev)) if ev else None
        allreduce["exposed_steps_timed"] = len(ev)
        runner.core.ar_events = None

    params_checksum = None
    if hasattr(runner, "core"):                     # outside the timed region: what the run left in the parameters (path-equivalence checks)
        params_checksum = int(runner.core.params.view(torch.int32).to(torch.int64).sum().item())
    if world > 1 or force_dist:
        ctypes.CDLL(None).fflush(None)              # every rank's buffered library output leaves before rank 0's line does
        sys.stdout.flush(); sys.stderr.flush()
        dist.barrier()
    if rank == 0:
        units = runner.units_per_step() if hasattr(runner, "units_per_step") else n_local
        out = {
            "metric": METRIC,
            "value": units * world * args.steps / dt,
            "unit": getattr(runner, "unit", "env_steps/s"),
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
        dg = importlib.import_module("deepq-decoding_amd._digest").csrc_digest()
        lib_dg = importlib.import_module("deepq-decoding_amd._lib").lib().dq_build_digest().decode()
        out["build"] = dict(library_digest=lib_dg[:16], sources_digest=dg[:16], library_built_from_these_sources=lib_dg == dg)      # (the .so is prebuilt: say which sources it is)
        if world == 1 and mode == "loop" and args.updates_per_step == 1 and args.ratio_steps > 0 and hasattr(runner, "reference_ratio_leg"):
            # the driver's line carries the whole claim: the same loop at the reference's training intensity (k = 32 n / B updates per vector step)
            out["reference_replay_ratio"] = runner.reference_ratio_leg(steps=args.ratio_steps, warmup=max(2, args.ratio_steps // 10))
        if params_checksum is not None:
            out["params_checksum"] = params_checksum
        if world > 1 or force_dist:
            out["rccl_ranks"] = rccl_ranks                            # (from ncclCommCount behind a checked pre-flight all-reduce, above)
            out["dist_backend"] = "rccl" if backend == "nccl" else backend
            out["replicas_identical"] = replicas_identical
            out["allreduce"] = allreduce
            out["dist_mode"] = dict(dist_tune, how="both forms timed before the timed region, the faster one ran") if dist_tune else \
                dict(chosen=os.environ.get("DQ_DIST_MODE", "single"), how="DQ_DIST_MODE" if "DQ_DIST_MODE" in os.environ else "default")
        if world == 1 and not args.no_cpu_baseline:
            full = dict(cfg, n_envs=n_local)
            if mode == "env":
                out["cpu_baseline"] = cpu_baseline_env(full)
            else:
                bl = importlib.import_module("deepq-decoding_amd.bench_loop")
                out["cpu_baseline"] = cpu_baseline_loop(full, runner.B, runner.eps, bl.C_LAYERS, bl.FF_LAYERS, mode=mode)
            if args.config in ("c1", "c2", "c3", "c5"):
                # the environment alone on one core and on all host cores (C oracle, a port of Environments.py)
                one = cpu_baseline_env(dict(CONFIGS[args.config], n_envs=4096), seconds=3.0)
                allc, cores = cpu_baseline_env_all_cores(args.config)
                out["cpu_baseline"].update(env_only_1core_steps_per_s=one["value"], env_only_all_cores_steps_per_s=allc, env_only_cores=cores, env_only_cgroup_cpu_quota_cores=cpu_quota_cores(),
                                           env_only_sample="C oracle environment + uniform-legal policy, 4096 lattices per process, one process per "
                                                           "host core, 5 s, lattice-steps summed")
        print(json.dumps(out), file=json_out, flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        if hasattr(runner, "core"):
            runner.core.close_comm()            # the learner's own communicator first, on every rank, everything drained
        dist.destroy_process_group()


def allreduce_probe(torch, dist, core, backend, iters=20):
    """The step's two gradient all-reduces on their own (untimed region): bytes and average microseconds, HIP events on the current
    stream around blocking-semantics collectives (the stream waits for the communicator's stream)."""
    nconv = core.net.n_conv_params
    parts = {"dense": core.grads[nconv:].clone(), "conv": core.grads[:nconv].clone()}
    mode = os.environ.get("DQ_DIST_MODE", "single")
    native = getattr(core, "_rccl", None) is not None
    out = {"backend": "rccl" if backend == "nccl" else backend,
           "per_step": ("one all-reduce of the whole flat gradient behind the backward, " + ("on the step's own stream through the learner's RCCL communicator"
                        if native else "through torch.distributed")) if mode == "single" else
                       ("dense range on a second stream through the learner's RCCL communicator while the convolutional backward runs, convolutional range "
                        "in-stream (DQ_DIST_MODE=overlap)") if mode == "overlap" and native else
                       "dense range asynchronous behind the convolutional backward, convolutional range on the critical path (DQ_DIST_MODE=split)",
           "in_stream_rccl": native}
    for name, buf in parts.items():
        for _ in range(3):
            dist.all_reduce(buf)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf)
        e1.record()
        torch.cuda.synchronize()
        out[name + "_bytes"] = buf.numel() * 4
        out[name + "_us"] = 1e3 * e0.elapsed_time(e1) / iters
    return out


class EnvOnly:
    """Environment kernel + device policy only (uniform over legal actions, auto-reset): T agent steps per LAUNCH (dq_env_act_steps: selection, step / reset, the
    transition into a replay ring whose slot advances on the device, the lattices' state in registers from step to step), so that the figure is the kernel's and
    not one launch latency per 4096-lattice step (SURVEY.md section 7 "hard parts")."""
    dtype = "u64 bit-planes / u8"

    def __init__(self, dq, cfg, n_local, rank, steps_per_launch=16, ring_slots=32, obs_form="patch", config_name="c3"):
        import torch
        self.torch = torch
        self.cfg, self.n, self.config_name = cfg, n_local, config_name
        self.env = dq.VectorEnv(n_envs=n_local, env_id_base=rank * n_local, **cfg)
        self.env.reset()
        self.T = max(1, int(steps_per_launch))
        self.slots = max(ring_slots, 2)
        dev = self.env.device
        self.action = torch.zeros((self.slots, n_local), dtype=torch.int32, device=dev)
        self.reward = torch.zeros((self.slots, n_local), dtype=torch.float32, device=dev)
        self.done = torch.zeros((self.slots, n_local), dtype=torch.uint8, device=dev)
        # the successor observation as the LOOP stores it -- patch words, d * d u32 per lattice -- or as the reference's uint8 planes
        self.obs_form = obs_form
        self.obs = torch.zeros((self.slots, n_local) + tuple(self.env.obs_shape), dtype=torch.uint8, device=dev) if obs_form == "uint8" else None
        self.patch = torch.zeros((self.slots, n_local, self.env.patch_stride), dtype=torch.int32, device=dev) if obs_form == "patch" else None
        self.t = 0
        self.pending = 0             # steps of the last launch not yet counted
        self.left = None             # steps left in the timed region (arm)
        self.bl = importlib.import_module("deepq-decoding_amd.bench_loop")
        self.L = importlib.import_module("deepq-decoding_amd._lib").lib()
        self.armed = False
        self.launch_steps = []

    def arm(self, steps):
        # a sample of the launches carries a HIP event pair (the dispatch's own timestamps, prof.hip)
        self.pending, self.left = 0, steps
        self.launch_steps = []
        self.armed = self.bl.prof_arm(self.L, "env_kernel", (steps + self.T - 1) // self.T, 1)

    def step(self, timed):
        """One vector step; every T-th call launches the next T (the timed region's last launch: what is left of it)."""
        if self.pending == 0:
            k = self.T if self.left is None else max(1, min(self.T, self.left))
            self.env.act_steps(k, self.t, self.action, self.reward, self.done, obs_ring=self.obs, patch_ring=self.patch, slot0=self.t % self.slots)
            self.t += k
            self.pending = k
            if self.left is not None:
                self.launch_steps.append(k)
        self.pending -= 1
        if self.left is not None:
            self.left -= 1

    def config(self):
        return dict(policy="uniform over legal actions (device, fused in front of the step)", auto_reset=True, steps_per_launch=self.T,
                    ring_slots=self.slots, observations=("patch words (d*d u32 per lattice)" if self.obs_form == "patch" else "uint8 planes") + " into a replay ring")

    def report(self, steps, dt, world):
        if not self.armed:
            return {"roofline": None}
        n, ms, lo, hi = ctypes.c_int(), ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        self.L.dq_prof_collect_spread(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(lo), ctypes.byref(hi))
        symbol = self.L.dq_prof_kernel_symbol(0).decode() or "env_kernel"
        self.L.dq_prof_arm(-1, 0)
        launches, total_ms = n.value, ms.value
        if not launches:
            return {"roofline": None}
        ms_l = total_ms / launches
        # B_env per lattice-step = 2 S (state record read + written) + scalars + legal words + the uint8 observation; with T steps per launch the
        # record moves once per LAUNCH, the rest once per step
        S = 96 if self.cfg["d"] <= 5 else 160
        per_step = env_bytes_per_step(self.cfg) - 2 * S
        if self.obs_form == "patch":     # d * d words instead of the C x (2d+1)^2 uint8 image
            layers = 1 if self.cfg["error_model"] == "X" else (3 if self.cfg["use_Y"] else 2)
            per_step += 4 * self.cfg["d"] ** 2 - (self.cfg["volume_depth"] + layers) * (2 * self.cfg["d"] + 1) ** 2
        k = sum(self.launch_steps) / max(1, len(self.launch_steps))          # steps per launch in the timed region
        bytes_per_launch = self.n * (per_step * k + 2 * S)
        achieved = bytes_per_launch / (ms_l * 1e-3) / 1e9
        return {"roofline": dict(kernel=symbol, family="env_kernel", bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s",
                                 frac=achieved / HBM_PEAK_GBS, traffic=self.bl.pmc_traffic(symbol, "env", self.config_name) if self.obs_form == "patch" and k == 64 else None, avg_launch_us=ms_l * 1e3,
                                 min_launch_us=lo.value * 1e3, max_launch_us=hi.value * 1e3, launches_timed=launches, steps_per_launch=k,
                                 algorithmic_bytes_per_launch=bytes_per_launch, lattice_steps_per_s_in_kernel=self.n * k / (ms_l * 1e-3))}


if __name__ == "__main__":
    main()
