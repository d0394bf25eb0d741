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


def exchange_unique_id(raw, err, rank, dev, group=None):
    """Rank 0's 128-byte ncclUniqueId to every rank through the process group.  Every rank takes part in BOTH collectives whatever happened
    to it before (a rank that raised in front of them would leave the others waiting in the broadcast), and nobody enters ncclCommInitRank --
    which blocks until all ranks have joined -- unless everybody is ready.  `raw`: the id's bytes on rank 0 (ALL 128 of them: the id is binary,
    a ctypes char-array field read as an attribute stops at its first zero byte), None elsewhere; `err`: what went wrong on this rank so far."""
    buf = bytearray(raw) if rank == 0 and raw is not None else bytearray(128)
    if len(buf) != 128:
        buf, err = bytearray(128), err or RuntimeError(f"ncclUniqueId of {len(buf)} bytes")
    t = torch.frombuffer(buf, dtype=torch.uint8).clone().to(dev)
    dist.broadcast(t, src=0, group=group)
    ready = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=dev)
    dist.all_reduce(ready, op=dist.ReduceOp.MIN, group=group)
    if int(ready.item()) == 0:
        raise RuntimeError(f"a rank could not prepare its RCCL communicator ({err if err is not None else 'another rank'})")
    return t.cpu().numpy().tobytes()


# ---- RCCL on the caller's stream ---------------------------------------------------------------------------------------------------
class RcclComm:
    """A RCCL communicator of this build's own (librccl through ctypes), used for the one collective of the learner's hot path: the in-place
    sum of the flat fp32 gradient, enqueued ON THE CALLER'S STREAM between the backward's final reduction and the Adam kernel -- one more
    launch in the step's stream, no host synchronisation, no hand-off to a communicator stream.  (torch.distributed's RCCL collectives run on
    the process group's internal stream: every call orders that stream behind the current one and the current one behind it again with
    events, and costs ~30 us of host time; measured on a one-rank group, DESIGN.md section 7: the step's several-GPU branch 226 us with the
    dense range all-reduced asynchronously through torch.distributed, 195-208 us with one synchronous torch.distributed all-reduce.)
    torch.distributed still does the rendezvous: rank 0's ncclUniqueId travels through the default process group."""

    class _UniqueId(__import__("ctypes").Structure):
        _fields_ = [("internal", __import__("ctypes").c_char * 128)]

    def __init__(self, rank, world, device, group=None):
        import ctypes
        self.rank, self.world, self.device = rank, world, torch.device(device)
        uid, err = self._UniqueId(), None
        try:
            libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
            path = os.path.join(libdir, "librccl.so")
            self.lib = ctypes.CDLL(path if os.path.exists(path) else "librccl.so")       # the library torch itself is linked against
            L = self.lib
            L.ncclGetErrorString.restype = ctypes.c_char_p
            L.ncclGetUniqueId.argtypes = [ctypes.POINTER(self._UniqueId)]
            L.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, self._UniqueId, ctypes.c_int]
            L.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
            L.ncclCommDestroy.argtypes = [ctypes.c_void_p]
            L.ncclCommAbort.argtypes = [ctypes.c_void_p]
            L.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
            if rank == 0:
                self._check(L.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        except (OSError, RuntimeError, AttributeError) as e:
            err = e
        if world > 1:
            dev = self.device if dist.get_backend(group) == "nccl" else torch.device("cpu")
            raw = exchange_unique_id(ctypes.string_at(ctypes.byref(uid), 128) if rank == 0 else None, err, rank, dev, group)
            ctypes.memmove(ctypes.byref(uid), raw, 128)
        elif err is not None:
            raise err
        self.comm = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what}: {self.lib.ncclGetErrorString(rc).decode()} (rank {self.rank})")

    def allreduce_sum_(self, flat):
        """In-place float32 sum over the ranks, enqueued on the current stream of `flat`'s device."""
        assert flat.dtype == torch.float32 and flat.is_cuda and flat.is_contiguous()
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        self._check(self.lib.ncclAllReduce(flat.data_ptr(), flat.data_ptr(), flat.numel(), 7, 0, self.comm, stream), "ncclAllReduce")   # ncclFloat32, ncclSum
        return flat

    def count(self):
        """Ranks of the communicator as RCCL itself reports them (ncclCommCount) -- not WORLD_SIZE: a mis-joined communicator shows here."""
        import ctypes
        n = ctypes.c_int(-1)
        self._check(self.lib.ncclCommCount(self.comm, ctypes.byref(n)), "ncclCommCount")
        return n.value

    def preflight(self):
        """One all-reduce of a rank-stamped vector, checked on the host against what `world` ranks must give: element i = sum_r (r + i / 8) over the ranks
        = world (world - 1) / 2 + world i / 8.  Raises on every rank that sees anything else -- a communicator that was joined by the wrong set of ranks
        (or with a truncated ncclUniqueId: the bug fixed in round 4) fails here, loudly, before any timed region.  Returns ncclCommCount's answer."""
        n = self.count()
        if n != self.world:
            raise RuntimeError(f"RCCL communicator holds {n} ranks, {self.world} expected (rank {self.rank})")
        i = torch.arange(64, dtype=torch.float32, device=self.device) / 8.0
        v = (float(self.rank) + i).contiguous()
        self.allreduce_sum_(v)
        torch.cuda.synchronize(self.device)
        want = self.world * (self.world - 1) / 2.0 + self.world * i
        if not torch.equal(v, want):
            raise RuntimeError(f"RCCL pre-flight all-reduce over {self.world} ranks returned {v[:4].tolist()} ..., expected {want[:4].tolist()} ... (rank {self.rank})")
        return n

    def close(self, abort=False):
        """ncclCommDestroy (every rank has issued the same collectives and drained them) -- or, leaving through an exception, ncclCommAbort: a destroy
        can wait for peers that are still inside a collective this rank will never join."""
        if getattr(self, "comm", None):
            (self.lib.ncclCommAbort if abort else self.lib.ncclCommDestroy)(self.comm)
            self.comm = None


def make_rccl(rank, world, device, group=None):
    """The learner's own communicator when the process group runs on RCCL (or DQ_DIST_FORCE drives the several-GPU branch on one GPU);
    None for gloo groups (CPU tests, DQ_DIST_BACKEND=gloo) and when DQ_DIST_NATIVE=0."""
    if os.environ.get("DQ_DIST_NATIVE", "1") == "0" or not (dist.is_available() and dist.is_initialized()):
        return None
    if dist.get_backend(group) != "nccl" or not torch.cuda.is_available():
        return None
    comm = None
    try:
        comm = RcclComm(rank, world, device, group)
    except (OSError, RuntimeError, AttributeError) as e:            # library or communicator unavailable here: the ranks agree on the fallback below
        import warnings
        warnings.warn(f"no RCCL communicator of our own ({e}); the gradient all-reduce goes through torch.distributed")
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device=device)
    if world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)     # every rank takes the same path
    if int(ok.item()) == 0:
        if comm is not None:
            comm.close()
        return None
    return comm
