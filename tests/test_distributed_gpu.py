"""Two ranks through the whole bench.py path on the ONE GPU of the test box (gloo moves the gradient; RCCL needs one GPU per rank):
the several-GPU branch of the learner -- TD step + dense backward, asynchronous all-reduce of the dense gradient behind the
convolutional backward, all-reduce of the convolutional range, separate Adam -- keeps the replicas bit-identical."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_stay_identical():
    env = dict(os.environ, DQ_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "3", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{"metric"')][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["replicas_identical"] is True and out["scaling"] == "weak"
    assert out["config"]["grad_allreduce"].startswith("RCCL") and out["value"] > 0
