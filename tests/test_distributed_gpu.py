"""Two ranks through the whole bench.py path on the ONE GPU of the test box (gloo moves the gradient; RCCL needs one GPU per rank):
the several-GPU branch of the learner -- TD step + dense backward, asynchronous all-reduce of the dense gradient behind the
convolutional backward, all-reduce of the convolutional range, separate Adam -- keeps the replicas bit-identical.  `python bench.py
--gpus 2` launches its own ranks (no torchrun wrapper); a world size that differs from --gpus is refused."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra)
    return env


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["single", "overlap", None])
def test_bench_self_launches_two_ranks_that_stay_identical(mode):
    # (overlap: on a gloo group -- no RCCL communicator of the learner's own -- the mode takes the split form through torch.distributed)
    # mode None (round 6): no DQ_DIST_MODE -- bench.py times both forms before the timed region, every rank picks the same one, the line says which and why
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--no-cpu-baseline", "--dist-probe-steps", "6"]
    env = _clean_env(DQ_DIST_BACKEND="gloo", **({"DQ_DIST_MODE": mode} if mode else {}))
    if mode is None:
        env.pop("DQ_DIST_MODE", None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(line) == 1                                           # rank 0 only
    assert r.stdout.strip().splitlines() == line, r.stdout[-2000:]  # ... and nothing else on stdout, from any rank
    out = json.loads(line[0])
    assert out["n_gpus"] == 2 and out["replicas_identical"] is True and out["scaling"] == "weak"
    assert out["config"]["grad_allreduce"].startswith("RCCL") and out["value"] > 0
    assert out["dist_backend"] == "gloo" and out["rccl_ranks"] == 0
    ar = out["allreduce"]
    assert ar["dense_bytes"] + ar["conv_bytes"] == 4 * out["config"]["n_params"] and ar["dense_us"] > 0 and ar["conv_us"] > 0
    dm = out["dist_mode"]
    if mode is None:
        assert dm["chosen"] in ("single", "overlap") and set(dm["us_per_step"]) == {"single", "overlap"} and dm["probe_steps"] == 6
        assert dm["us_per_step"][dm["chosen"]] == min(dm["us_per_step"].values()) and ar["exposed_us_per_step"] is not None
    else:
        assert dm["chosen"] == mode and dm["how"] == "DQ_DIST_MODE"


@pytest.mark.gpu
def test_one_rank_rccl_group_drives_the_several_gpu_branch():
    """The several-GPU learner branch through the REAL backend on the one-GPU box: DQ_DIST_FORCE=1 makes bench.py create a one-rank RCCL
    process group and DQNCore take its several-GPU branch -- split backward without the fused optimizer step, the flat gradient summed over the
    ranks, separate Adam launch -- in each of its three forms: the default (ONE all-reduce on the step's own stream through the learner's own
    RCCL communicator, dist.RcclComm), the same through torch.distributed (DQ_DIST_NATIVE=0), and the split form (DQ_DIST_MODE=split: dense
    range asynchronously on the process group's stream behind the convolutional backward, convolutional range on the critical path).  With
    one rank the sum is the identity, so the parameters after the run must be the bits the one-GPU branch (Adam fused on the reduction) leaves."""
    def run(**extra):
        # (--dist-probe-steps 0: no timing of both exchange forms in front of the timed region -- its extra steps would move the parameters this test compares)
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "24", "--warmup", "4", "--no-cpu-baseline", "--dist-probe-steps", "0", "--ratio-steps", "0"]
        r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(**extra), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        lines = r.stdout.strip().splitlines()
        # stdout is the ONE contract line, also with a RCCL communicator in the process (its version banner, C stdio flushed at exit, goes to stderr)
        assert len(lines) == 1 and lines[0].startswith('{"metric"'), r.stdout[-2000:]
        return json.loads(lines[0])
    plain = run()
    assert "rccl_ranks" not in plain
    for extra, native, word in ((dict(), True, "own stream"), (dict(DQ_DIST_NATIVE="0"), False, "torch.distributed"), (dict(DQ_DIST_MODE="split"), False, "asynchronous"),
                                (dict(DQ_DIST_MODE="overlap"), True, "second stream")):
        forced = run(DQ_DIST_FORCE="1", **extra)
        assert forced["rccl_ranks"] == 1 and forced["dist_backend"] == "rccl" and forced["replicas_identical"] is True
        ar = forced["allreduce"]
        assert ar["backend"] == "rccl" and ar["dense_us"] > 0 and ar["conv_us"] > 0 and ar["exposed_us_per_step"] > 0
        assert ar["in_stream_rccl"] is native and word in ar["per_step"], ar
        assert forced["params_checksum"] == plain["params_checksum"], extra


@pytest.mark.gpu
def test_bench_refuses_a_world_that_differs_from_gpus():
    # one process that claims to be a 1-rank world while --gpus says 2: no JSON line, non-zero exit
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       cwd=ROOT, env=_clean_env(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '{"metric"' not in r.stdout and "refusing" in r.stderr
    # RCCL needs one GPU per rank: two RCCL ranks on the one-GPU box are refused rather than silently sharing the device
    import torch
    if torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                           cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and '{"metric"' not in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["act", "learn", "env"])
def test_bench_modes_print_one_contract_line(mode):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", mode, "--steps", "20", "--warmup", "3", "--no-cpu-baseline"],
                       cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["unit"] == ("dqn_samples/s" if mode == "learn" else "env_steps/s")
    assert f"mode={mode}" in out["config"]["workload"]


@pytest.mark.gpu
def test_fit_stops_on_the_same_step_on_every_rank(tmp_path):
    """DQNAgent.fit under two ranks whose local lifetimes differ by orders of magnitude: the early-stopping decision comes from the
    all-reduced episode statistics, so both leave on the same step with bit-identical parameters (a rank leaving alone would hang the
    other in the gradient all-reduce -- the subprocess timeout catches that); FileLogger output comes from rank 0 only."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "_fit_two_ranks.py"), str(tmp_path)]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1))
    assert a == b, (a, b)
    assert a["stopped"] is True and a["step"] < 64 * 400 and a["updates"] > 0
    data = json.load(open(tmp_path / "training_history.json"))
    assert len(data["episode"]) == a["records"]


@pytest.mark.gpu
def test_single_lattice_fit_under_two_ranks_counts_every_step(tmp_path):
    """N = 1 per rank, two ranks: the extra auto-reset step behind an episode end is taken only by the rank whose own lattice finished
    (the noisy rank 0: one extra launch per local episode; the nearly noiseless rank 1: none), both ranks take the same number of
    collective updates and hold identical parameters."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29543", os.path.join(ROOT, "tests", "_fit_two_ranks.py"), str(tmp_path), "single"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1))
    assert a["step"] == b["step"] == 300 and a["updates"] == b["updates"] > 0 and a["params"] == b["params"]
    # every episode end costs exactly ONE uncounted reset step, on the rank that owns the lattice (before the fix both ranks spent one
    # whenever either finished: extra_a == extra_b == number of episode ends)
    extra_a, extra_b = a["vector_steps"] - 300, b["vector_steps"] - 300
    assert a["episodes_global"] == b["episodes_global"] == extra_a + extra_b and extra_a > extra_b > 0


@pytest.mark.gpu
def test_single_lattice_fit_under_two_ranks_with_a_longer_sync_interval(tmp_path):
    """The same with sync_interval = 4: the ranks' environment-launch counters differ (one uncounted reset step per LOCAL episode end), the
    synchronisation points -- statistics all-reduce, collective update -- are gated on the counted steps of this fit(), the same on every
    rank: no deadlock (the subprocess timeout would catch it), same number of updates, identical parameters."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29545", os.path.join(ROOT, "tests", "_fit_two_ranks.py"), str(tmp_path), "single4"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1))
    assert a["step"] == b["step"] == 300 and a["updates"] == b["updates"] > 0 and a["params"] == b["params"]
    assert a["vector_steps"] != b["vector_steps"]            # (the situation the gate has to survive)


@pytest.mark.gpu
def test_range_guard_discards_the_update_on_every_rank(tmp_path):
    """Two ranks (gloo, one GPU), one of them with TD errors of ~1e6 in its minibatch: no parameter moves on either rank, both raise DQ_ERR_RANGE at
    their next read_metrics(), the replicas stay bit-identical and the loop carries on afterwards."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(ROOT, "tests", "_fit_two_ranks.py"), str(tmp_path), "range"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(DQ_TD_AUTOSCALE="0"), capture_output=True, text=True, timeout=600)      # (0: the host-known scale for good -- the error is raised)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1))
    assert a == b, (a, b)
    assert a["unchanged"] and a["raised"] and a["finite"] and a["moved"] and a["discarded"] == 1 and not a["auto_scale"]


@pytest.mark.gpu
def test_range_guard_default_path_keeps_two_ranks_identical_and_running(tmp_path):
    """The DEFAULT settings (DQ_TD_AUTOSCALE unset): the same situation does not raise -- both ranks discard the update whole, warn at the same
    synchronisation, count ONE discarded update each (dq_qnet_range_discarded: rank 1 learns of it through the NaNs of the all-reduced gradient), switch to
    the measured gradient scale, and then carry rank 0's poisoned memory with finite, bit-identical parameters."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29553", os.path.join(ROOT, "tests", "_fit_two_ranks.py"), str(tmp_path), "rangeauto"]
    env = _clean_env()
    env.pop("DQ_TD_AUTOSCALE", None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1))
    assert a == b, (a, b)
    assert a["unchanged"] and not a["raised"] and a["warned"] and a["auto_scale"] and a["discarded"] == 1 and a["finite"] and a["moved"]


@pytest.mark.gpu
def test_a_range_error_inside_fit_leaves_every_rank_with_the_communicator_closed(tmp_path):
    """DQNAgent.fit's exits: a DQ_ERR_RANGE raised inside fit() -- rank 0's replay rewards poisoned mid-run -- reaches EVERY rank at
    the same synchronisation; each leaves fit() through its exception handler with the learner's communicator closed (aborted, not destroyed), the
    agent no longer `training`, the replicas still finite and identical, and the process group usable.  Two ranks over gloo, then ONE rank through
    the real RCCL backend (DQ_DIST_FORCE=1: the learner holds a RcclComm of its own there, so `closed` means ncclCommAbort has run)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(ROOT, "tests", "_fit_two_ranks.py"), str(tmp_path), "fitrange"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(DQ_TD_AUTOSCALE="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = (json.load(open(tmp_path / f"rank{k}.json")) for k in (0, 1))
    for x in (a, b):
        assert x["raised"] and x["status"] == -6 and x["closed"] and x["finite"] and not x["training"] and x["group_sum"] == 3.0, x
    assert a["step"] == b["step"] and a["params"] == b["params"]
    one = tmp_path / "one"
    one.mkdir()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29551", os.path.join(ROOT, "tests", "_fit_two_ranks.py"), str(one), "fitrange"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(DQ_DIST_FORCE="1", DQ_TD_AUTOSCALE="0"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    x = json.load(open(one / "rank0.json"))
    assert x["raised"] and x["status"] == -6 and x["closed"] and x["finite"] and not x["training"] and x["group_sum"] == 1.0, x


@pytest.mark.gpu
def test_rccl_preflight_on_a_one_rank_group():
    """dist.RcclComm.preflight (bench.py --gpus N runs it before its warm-up): ncclCommCount's answer and a checked all-reduce of a rank-stamped vector."""
    code = ("import os, torch, importlib, torch.distributed as dist\n"
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29553', RANK='0', WORLD_SIZE='1')\n"
            "torch.cuda.set_device(0); dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
            "D = importlib.import_module('deepq-decoding_amd.dist')\n"
            "c = D.RcclComm(0, 1, 'cuda:0')\n"
            "assert c.count() == 1 and c.preflight() == 1\n"
            "c.world = 2\n"
            "try:\n    c.preflight(); raise SystemExit('a one-rank communicator passed a two-rank pre-flight')\n"
            "except RuntimeError as e:\n    assert 'holds 1 ranks, 2 expected' in str(e)\n"
            "c.close(); dist.destroy_process_group(); print('ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout.splitlines(), r.stderr[-2000:]     # (RCCL prints its version banner through C stdio at exit, behind the line)


@pytest.mark.gpu
def test_default_bench_line_carries_the_whole_claim():
    """a driver-shaped run (`python bench.py --steps 20 --warmup 5`, cpu baseline off here for time) prints ONE line whose `roofline` names the
    kernel SYMBOL that ran beside its family, carries the launch spread and a non-null PMC traffic (the committed pass matches the sources: the CPU suite's
    test_pmc_stamp_matches_sources), and whose `reference_replay_ratio` is a second timed leg at 32 trained samples per environment step."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--ratio-steps", "20"]
    r = subprocess.run(cmd, cwd=ROOT, env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["warmup"] == 5 and out["vs_baseline"] is None and out["higher_is_better"] is True
    assert "d=5" in out["config"]["workload"] and "4096 lattices" in out["config"]["workload"] and out["unit"] == "env_steps/s"
    roof = out["roofline"]
    assert roof["kernel"] == "conv_wave_kernel" and roof["family"] == "conv_chain_kernel" and roof["bound"] == "mfma"
    assert 0.0 < roof["frac"] < 1.0 and roof["min_launch_us"] <= roof["avg_launch_us"] <= roof["max_launch_us"] and roof["launches_timed"] == 5
    assert roof["traffic"] is not None and 1e6 < roof["traffic"] < 2e8, "profiles/pmc_traffic_loop_c3.json is stale or does not list conv_wave_kernel"
    # the dominant kernel's live duration fits the step, the algorithmic rate stays below the pipe's peak
    assert roof["avg_launch_us"] < 1e3 * out["ms_per_step"] and roof["achieved"] < roof["peak"]
    ratio = out["reference_replay_ratio"]
    assert ratio["updates_per_vector_step"] == 32 and ratio["samples_trained_per_env_step"] == 32.0 and ratio["steps"] == 20
    assert ratio["value"] > 5e5 and abs(ratio["value"] - 4096 * 1e3 / ratio["ms_per_step"]) < 1.0
