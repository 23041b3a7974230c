"""Cross-replica communicators for the data-parallel trainer.

The reference sums the clones' gradients with `tf.add_n` on the CPU (slim/deployment/model_deploy.py:
414-444) and shares one CPU copy of every variable between towers (:640-675). Here every rank holds a
replica in HBM and the product path is RCCL behind the library's C ABI (`mtlssl_comm_*`,
include/mtlssl_hip.h): `RcclComm`. `torch.distributed` (gloo) is only the host-side channel that
ships the 128-byte RCCL unique id from rank 0 to the other ranks, and — as `GlooComm` — the stand-in
transport of the CPU unit tests and of the two-replicas-on-one-GPU test (RCCL needs one device per
rank).
"""
import contextlib
import ctypes
import os
import sys

import torch

from .lib import lib, ptr

_DTYPES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3}
_OPS = {"sum": 0, "max": 1, "min": 2}
ID_BYTES = 128


def _stream_ptr(stream):
    return (stream if stream is not None else torch.cuda.current_stream()).cuda_stream


@contextlib.contextmanager
def _stdout_to_stderr():
    """RCCL prints a version banner on C stdout while it initialises; callers that own stdout (bench.py prints
    exactly one JSON line there) must not see it — send fd 1 to fd 2 for the duration, flushing C stdio inside."""
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    saved = os.dup(1)
    os.dup2(2, 1)
    try:
        yield
    finally:
        libc.fflush(None)
        os.dup2(saved, 1)
        os.close(saved)


def _apply_rccl_knobs():
    """Knobs for sharing the chip between the step's compute streams and RCCL's own kernels (each RCCL channel is a
    persistent workgroup that holds a CU for the duration of a collective): MTLSSL_COMM_MAX_CHANNELS /
    MTLSSL_COMM_MIN_CHANNELS are handed to RCCL as NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS (read by RCCL when the
    communicator is created; an explicit NCCL_* setting in the environment wins). tools/cu_thief_probe.py sizes what a
    given channel count costs the step on one GPU."""
    for mine, theirs in (("MTLSSL_COMM_MAX_CHANNELS", "NCCL_MAX_NCHANNELS"), ("MTLSSL_COMM_MIN_CHANNELS", "NCCL_MIN_NCHANNELS")):
        v = os.environ.get(mine)
        if v and theirs not in os.environ:
            os.environ[theirs] = str(int(v))


def comm_stream_priority():
    """Priority of the gradient all-reduce stream (MTLSSL_COMM_STREAM_PRIORITY: 0 default, -1 high). The collectives
    are few, large and off the critical path until the optimizer waits for them; high priority lets their channels
    claim CUs at kernel boundaries instead of queueing behind three saturated compute streams."""
    return int(os.environ.get("MTLSSL_COMM_STREAM_PRIORITY", "0"))


def _dist_agree(failed):
    """True on every rank iff `failed` is False on every rank (host channel: the initialised torch.distributed group;
    all_gather_object works on every backend — a CPU tensor all-reduce would not on an nccl-only group)."""
    import torch.distributed as dist
    flags = [None] * dist.get_world_size()
    dist.all_gather_object(flags, bool(failed))
    return not any(flags)


class RcclComm:
    """One RCCL communicator rank bound to `device` (mtlssl_comm_init). `exchange(id_or_None)` must
    return rank 0's unique id on every rank; the default uses the initialised torch.distributed
    group (any backend: object collectives only) as the host channel. world == 1 needs no channel at all. `agree(failed) -> bool` is the
    all-ranks AND of "my preconditions hold" over the same channel: the ranks settle it BEFORE the collective
    mtlssl_comm_init, so a rank that cannot load RCCL or see its device makes every rank raise instead of leaving the
    healthy ones blocked inside ncclCommInitRank."""
    backend = "rccl"

    def __init__(self, device, rank=0, world=1, exchange=None, agree=None):
        self.device = torch.device(device)
        assert self.device.type == "cuda", "RCCL communicators live on a GPU"
        self._lib = lib()
        _apply_rccl_knobs()
        pre = None
        try:
            if self.device.index is not None and self.device.index >= torch.cuda.device_count():
                raise RuntimeError("device %s is not visible (%d devices)" % (self.device, torch.cuda.device_count()))
            with torch.cuda.device(self.device):
                self._lib.comm_available(None, None)
        except Exception as e:
            pre = e
        if world > 1:
            if agree is None and exchange is not None:
                # a caller with its own host channel and no agree step: torch.distributed may not even be initialised.
                # Without a way to settle it across ranks a failed precondition raises here, on this rank, before the
                # collective init (the other ranks then fail in `exchange` / mtlssl_comm_init on their side).
                if pre is not None:
                    raise pre
            elif not (agree or _dist_agree)(pre is not None):
                raise RuntimeError("RCCL preconditions failed on at least one rank%s" % (": %r" % pre if pre else ""))
        elif pre is not None:
            raise pre
        uid = ctypes.create_string_buffer(ID_BYTES)
        err = None
        if rank == 0:
            try:
                self._lib.comm_unique_id(uid)
            except Exception as e:        # e.g. librccl.so cannot be loaded: the other ranks must not wait for an id
                err = e
        if world > 1:
            raw = (exchange or _dist_exchange)((b"" if err is not None else uid.raw) if rank == 0 else None)
            if not raw:
                raise RuntimeError("rank 0 could not create an RCCL unique id%s" % (": %r" % err if err else ""))
            uid = ctypes.create_string_buffer(raw, ID_BYTES)
        elif err is not None:
            raise err
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device), _stdout_to_stderr():
            self._lib.comm_init(ctypes.byref(handle), uid, world, rank)
        self._h = handle
        n, r, d, v = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        self._lib.comm_info(self._h, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d), ctypes.byref(v))
        # what RCCL reports, not what was asked for
        self.world, self.rank, self.rccl_device, self.rccl_version = n.value, r.value, d.value, v.value
        if (self.world, self.rank) != (world, rank):
            raise RuntimeError("RCCL reports rank %d of %d, expected %d of %d" % (self.rank, self.world, rank, world))

    def allreduce(self, t, op="sum", stream=None):
        """In-place all-reduce of a contiguous device tensor, asynchronous on `stream`."""
        assert t.is_cuda and t.is_contiguous()
        self._lib.comm_allreduce(self._h, ptr(t), t.numel(), _DTYPES[t.dtype], _OPS[op], _stream_ptr(stream))

    def broadcast(self, t, root=0, stream=None):
        assert t.is_cuda and t.is_contiguous()
        self._lib.comm_broadcast(self._h, ptr(t), t.numel() * t.element_size(), root, _stream_ptr(stream))

    def info(self):
        return {"backend": "rccl", "ranks": self.world, "rank": self.rank, "device": self.rccl_device,
                "rccl_version": self.rccl_version}

    def close(self):
        if self._h is not None and self._h.value:
            self._lib.comm_destroy(self._h)
        self._h = None


class GlooComm:
    """torch.distributed behind the same interface: the stand-in transport of the tests (default group, gloo), and —
    with a process group of backend 'nccl', i.e. RCCL through torch's binding — the fallback `default_comm` takes
    when the library's own wrappers cannot be initialised."""
    backend = "gloo"

    def __init__(self, group=None, label=None):
        import torch.distributed as dist
        assert dist.is_initialized()
        self._dist = dist
        self._group = group
        self._label = label
        self.backend = str(dist.get_backend(group))          # the group's actual backend (nccl for the fallback)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def _on(self, stream, fn):
        if stream is None:
            return fn()
        with torch.cuda.stream(stream):
            return fn()

    def allreduce(self, t, op="sum", stream=None):
        ops = {"sum": self._dist.ReduceOp.SUM, "max": self._dist.ReduceOp.MAX, "min": self._dist.ReduceOp.MIN}
        self._on(stream, lambda: self._dist.all_reduce(t, op=ops[op], group=self._group))

    def broadcast(self, t, root=0, stream=None):
        self._on(stream, lambda: self._dist.broadcast(t, root, group=self._group))

    def info(self):
        return {"backend": self._label or self._dist.get_backend(self._group), "ranks": self.world, "rank": self.rank}

    def close(self):
        pass


def _dist_exchange(uid):
    import torch.distributed as dist
    assert dist.is_initialized(), "RcclComm(world > 1) needs a host channel: initialise torch.distributed (gloo) first"
    box = [uid]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def default_comm(device):
    """The communicator a trainer uses when none is given: none for a single process, RCCL for a GPU
    replica of an initialised torch.distributed job (MTLSSL_DIST_BACKEND=gloo forces the stand-in, which
    lets the multi-rank code path run on one GPU), gloo for CPU tensors."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    dev = torch.device(device)
    if dev.type == "cuda" and os.environ.get("MTLSSL_DIST_BACKEND", "rccl") != "gloo":
        comm, err = None, None
        try:
            comm = RcclComm(dev, dist.get_rank(), dist.get_world_size())
        except Exception as e:
            err = e
        # every rank has to take the same branch: agree over the host channel
        failed = torch.tensor([0 if comm is not None else 1], dtype=torch.int32)
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed.item()) == 0:
            return comm
        if comm is not None:
            comm.close()
        sys.stderr.write("mtl_ssl_amd.comm: mtlssl_comm_init failed on at least one rank (%r here); falling back to "
                         "torch.distributed's nccl backend (RCCL through torch's binding)\n" % (err,))
        return GlooComm(group=dist.new_group(backend=_FALLBACK_BACKEND), label="torch-%s (fallback)" % _FALLBACK_BACKEND)
    return GlooComm()


_FALLBACK_BACKEND = "nccl"      # tests replace it with "gloo" to walk the fallback without a second GPU
