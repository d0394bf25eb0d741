"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU for tests).

The reference has no multi-device path at all; the sharding below is this build's design:
lattices are independent, so rank r owns global lattice ids [r*n_local, (r+1)*n_local) and its own replay
shard; the only coupling is the shared Q-network, kept identical on every rank by ONE all-reduce (sum) of the flat
fp32 gradient per optimizer step followed by the same Adam update everywhere.  Each rank scales its loss gradient by
1/(B_local*world) so the sum is the gradient of the global-minibatch mean.
"""
import os

import torch
import torch.distributed as dist


def env_int(name, default):
    return int(os.environ.get(name, default))


def init_from_env(backend=None):
    """Initialise the default process group from RANK/WORLD_SIZE/LOCAL_RANK/MASTER_* (torchrun).  Returns
    (rank, world, local_rank).  No-op for WORLD_SIZE == 1."""
    world, rank, local = env_int("WORLD_SIZE", 1), env_int("RANK", 0), env_int("LOCAL_RANK", 0)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def _active(group=None):
    """A collective has something to do: more than one rank -- or DQ_DIST_FORCE=1 with an initialised one-rank group, which drives the
    several-GPU code path (RCCL communicator, asynchronous all-reduce on the communicator's stream, split backward, separate Adam)
    through the real backend on a one-GPU box (tests/test_distributed_gpu.py; the collective itself is then a copy onto itself)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or os.environ.get("DQ_DIST_FORCE") == "1"


def dist_path(world_size):
    """True when the learner takes its several-GPU branch (DQNCore._learn)."""
    return world_size > 1 or (os.environ.get("DQ_DIST_FORCE") == "1" and dist.is_available() and dist.is_initialized())


def shard(rank, n_local, batch_local):
    """(env_id_base, sample_base) of a rank: global lattice ids and global minibatch-sample ids are contiguous per rank,
    so an R-rank run reproduces the lattices / dropout masks / replay draws of a 1-rank run of R*n_local lattices."""
    return rank * n_local, rank * batch_local


def grad_scale(batch_local, world):
    return 1.0 / (batch_local * world)


def allreduce_sum_(flat, group=None):
    """In-place sum of the flat gradient over all ranks (RCCL ring/tree over xGMI on the GPU box; 0.77 MB at c3)."""
    if _active(group):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def allreduce_sum_async(flat, group=None):
    """Starts the in-place sum and returns the work handle (None when there is nothing to do): the collective runs on the
    communicator's stream, ordered after the work already queued on the current stream, while later kernels on the current
    stream proceed; `handle.wait()` orders the current stream after it."""
    if _active(group):
        return dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=True)
    return None


def broadcast_(flat, src=0, group=None):
    if _active(group):
        dist.broadcast(flat, src=src, group=group)
    return flat
