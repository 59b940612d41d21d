"""Exchange step of the column-sharded fit.

Feature columns are split over ranks; every per-iteration quantity that is a sum over
features (un-normalised block scores t~, ||w~_b||^2, ||p_b||^2, X-sized sums of squares,
X W products, the Gram matrix) is all-reduced.  One process per GPU; the collective is
the C ABI's own RCCL communicator (``RcclComm``: ncclAllReduce on the context's stream, no host
synchronisation), or ``torch.distributed`` (``TorchDistComm``: backend "nccl" == RCCL on ROCm, "gloo" for the
CPU tests), where torch owns the exchange buffer and the communicator, nothing else.
"""
from __future__ import annotations

import numpy as np


class LocalComm:
    """Single device: no exchange."""
    rank = 0
    world_size = 1
    is_distributed = False
    device_views_ok = True       # allreduce_device accepts any DeviceArray (views of larger buffers included)

    def exchange_buffer(self, ctx, count):
        from ._capi import DeviceArray
        return DeviceArray(ctx, (count,))

    def allreduce_device(self, buf):
        return buf

    def allreduce_host(self, arr):
        return arr


class _TorchExchange:
    """Device buffer owned by torch so RCCL can reduce it in place; quacks like DeviceArray."""

    def __init__(self, tensor):
        self.tensor = tensor
        self.ptr = tensor.data_ptr()
        self.size = tensor.numel()
        self.nbytes = 8 * self.size
        self.shape = (self.size,)

    def get(self):
        return self.tensor.cpu().numpy()

    def set(self, arr):
        import torch
        self.tensor.copy_(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float64).ravel()))
        return self


class TorchDistComm:
    """All-reduce over a torch.distributed process group.

    device: torch.device of this rank's GPU ("cuda:<local_rank>") for the nccl/RCCL backend,
    or None/"cpu" for gloo (host tensors; used by the world_size-2 CPU tests).  A GPU device together with
    the gloo backend stages the exchange through the host: that is how several ranks can share ONE GPU
    (RCCL refuses duplicate devices), used to exercise the real engine's sharded path with world_size 2 on
    a single-GPU box."""
    is_distributed = True
    device_views_ok = False      # reduces torch tensors it allocated itself (exchange_buffer)

    def __init__(self, group=None, device=None):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self._torch, self._dist = torch, dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world_size = dist.get_world_size(group)
        self.device = torch.device(device) if device is not None else torch.device("cpu")
        self._host_staged = self.device.type != "cpu" and str(dist.get_backend(group)) == "gloo"
        if self.device.type != "cpu":
            # torch's caching allocator and RCCL share this device: keep the library's pool of freed buffers small
            try:
                from .runtime import get_context
                get_context().set_pool_cap(0.25)
            except Exception:
                pass

    def exchange_buffer(self, ctx, count):
        t = self._torch.zeros(int(count), dtype=self._torch.float64, device=self.device)
        return _TorchExchange(t)

    def allreduce_device(self, buf):
        # kernels ran on the null stream == torch's default stream, so ordering is implicit
        if self._host_staged:
            h = buf.tensor.cpu()
            self._dist.all_reduce(h, op=self._dist.ReduceOp.SUM, group=self.group)
            buf.tensor.copy_(h)
        else:
            self._dist.all_reduce(buf.tensor, op=self._dist.ReduceOp.SUM, group=self.group)
        return buf

    def allreduce_host(self, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64)
        t = self._torch.from_numpy(a.reshape(-1).copy())
        if self.device.type != "cpu" and not self._host_staged:
            t = t.to(self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM, group=self.group)
        return t.cpu().numpy().reshape(a.shape)


class RcclComm:
    """All-reduce through the C ABI's own RCCL communicator (csrc/rccl_comm.hip): the collective is enqueued on the
    context's stream, in place on library-owned device memory -- no torch tensor, no host synchronisation.  This is
    what lets a column-sharded one-response fit run as ONE device queue (pass, all-reduce, step kernel, pass, ...).

    ``bootstrap(payload)``: a callable that returns rank 0's ``payload`` (bytes) on every rank -- e.g. a broadcast
    over torch.distributed's store, MPI, or a file.  ``RcclComm.from_torch`` builds one from an initialised
    torch.distributed process group (any backend; only the 128-byte id goes through it)."""
    is_distributed = True
    device_views_ok = True

    def __init__(self, ctx, rank, world_size, bootstrap):
        import ctypes as C
        from . import _capi
        self.ctx, self.rank, self.world_size = ctx, int(rank), int(world_size)
        buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
        if self.rank == 0:
            _capi.check(ctx.lib.mbpls_hip_comm_unique_id(buf))
        payload = bootstrap(bytes(buf.raw) if self.rank == 0 else None)
        assert isinstance(payload, (bytes, bytearray)) and len(payload) == _capi.COMM_ID_BYTES
        idbuf = C.create_string_buffer(bytes(payload), _capi.COMM_ID_BYTES)
        _capi.check(ctx.lib.mbpls_hip_comm_init_rank(ctx.handle, self.world_size, self.rank, idbuf))
        self._small = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    @classmethod
    def from_torch(cls, ctx, group=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)

        # (broadcast_object_list takes a GLOBAL rank: the group's first member, not necessarily global rank 0)
        src = dist.get_global_rank(group, 0) if group is not None else 0

        def bootstrap(payload):
            box = [payload]
            dist.broadcast_object_list(box, src=src, group=group)
            return box[0]

        return cls(ctx, rank, world, bootstrap)

    def info(self):
        import ctypes as C
        w, r, v = C.c_int(0), C.c_int(0), C.c_int(0)
        from . import _capi
        _capi.check(self.ctx.lib.mbpls_hip_comm_info(self.ctx.handle, C.byref(w), C.byref(r), C.byref(v)))
        # ... and what the communicator itself says (ncclCommCount / ncclCommUserRank / ncclCommCuDevice)
        cn, cr, cd = C.c_int(-1), C.c_int(-1), C.c_int(-1)
        _capi.check(self.ctx.lib.mbpls_hip_comm_query(self.ctx.handle, C.byref(cn), C.byref(cr), C.byref(cd)))
        return dict(world_size=w.value, rank=r.value, rccl_version=v.value, rccl_ranks=cn.value,
                    rccl_user_rank=cr.value, rccl_device=cd.value)

    def close(self):
        if self.ctx is not None and self.ctx.handle:
            self.ctx.lib.mbpls_hip_comm_destroy(self.ctx.handle)
        self.ctx = None

    def set_timeout(self, seconds):
        """Deadline of one wait for the compute stream while collectives may be in flight (default 600 s).  A peer that
        dies makes RCCL report an asynchronous error within milliseconds; the deadline is for the peer that is alive
        but never arrives.  Either way the wait raises ``CommLost`` instead of hanging (csrc/rccl_comm.hip)."""
        from . import _capi
        _capi.check(self.ctx.lib.mbpls_hip_comm_set_timeout(self.ctx.handle, max(1, int(round(1e3 * seconds)))))

    def inject_fault(self, mode, polls=0):
        """test hook: the next all-reduce behaves like a collective whose peer has died (mode 1: RCCL's asynchronous
        error turns up after ``polls`` queries; mode 2: it never does -- the deadline has to catch it; 0: off)"""
        from . import _capi
        _capi.check(self.ctx.lib.mbpls_hip_comm_inject_fault(self.ctx.handle, int(mode), int(polls)))

    def exchange_buffer(self, ctx, count):
        from ._capi import DeviceArray
        return DeviceArray(ctx, (int(count),), zero=True)

    def allreduce_device(self, buf):
        from . import _capi
        _capi.check(self.ctx.lib.mbpls_hip_allreduce(self.ctx.handle, buf.ptr, buf.size))
        return buf

    def allreduce_host(self, arr):
        """small host arrays (block sizes, K x K cross products): staged through a device buffer"""
        a = np.ascontiguousarray(arr, dtype=np.float64)
        if a.size == 0:
            return a
        if self._small is None or self._small.size < a.size:
            if self._small is not None:
                self._small.free()
            self._small = self.ctx.empty(max(a.size, 4096))
        d = self._small
        from . import _capi
        _capi.check(self.ctx.lib.mbpls_hip_memcpy_h2d(self.ctx.handle, d.ptr, a.ctypes.data, a.nbytes))
        _capi.check(self.ctx.lib.mbpls_hip_allreduce(self.ctx.handle, d.ptr, a.size))
        out = np.empty_like(a)
        _capi.check(self.ctx.lib.mbpls_hip_memcpy_d2h(self.ctx.handle, out.ctypes.data, d.ptr, a.nbytes))
        return out


def shard_columns(block_sizes, rank, world_size):
    """Column range of every block owned by ``rank``: each block is split evenly so all ranks
    hold the same share of every block (balanced, per-block partials exist everywhere)."""
    out = []
    for pb in block_sizes:
        base, rem = divmod(int(pb), world_size)
        lo = rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        out.append((lo, hi))
    return out
