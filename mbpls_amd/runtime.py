"""Process-wide device context and (optional) column-sharding communicator."""
from __future__ import annotations

import contextlib

from . import _capi
from .comm import LocalComm

_contexts = {}
_state = {"device": 0, "comm": LocalComm()}


def set_device(device):
    _state["device"] = int(device)


_ctx_lock = __import__("threading").Lock()


def get_context(device=None):
    """The shared ``Context`` of a device (created on first use; raises if no GPU / no library)."""
    d = _state["device"] if device is None else int(device)
    ctx = _contexts.get(d)
    if ctx is None or ctx.handle is None:
        with _ctx_lock:                       # (two threads asking first at the same time: one context, not two)
            ctx = _contexts.get(d)
            if ctx is None or ctx.handle is None:
                ctx = _capi.Context(d)
                _contexts[d] = ctx
                _warm_up(ctx)
    return ctx


def _warm_up(ctx):
    """First-use costs of a process, paid when the context is created -- while the caller has not yet uploaded or
    generated its data -- instead of inside its first fit (measured at BASELINE config[1]: first fit 0.137 s against
    0.105 s warm before, 0.115 s with this): the kernels' code objects (loaded lazily, one per translation unit), the
    pinned bounce buffers, the copier thread (csrc/copier.hip), the BLAS thread-pool scan of threadpoolctl.  Synchronous on purpose: from
    a helper thread the code-object loads deadlocked against a dlopen of the main thread under rocprofv3 (the profiler
    hooks both).  Everything here is also done on demand; MBPLS_NO_WARMUP=1 switches it off."""
    import os
    if os.environ.get("MBPLS_NO_WARMUP") == "1":
        return
    try:
        ctx.lib.mbpls_hip_warmup(ctx.handle)
        _thread_limits()
    except Exception:          # a warm-up, never an error
        pass


def get_comm():
    return _state["comm"]


def set_comm(comm):
    _state["comm"] = comm if comm is not None else LocalComm()


@contextlib.contextmanager
def column_sharded(comm):
    """``with column_sharded(TorchDistComm(...)):`` -- inside, ``MBPLS.fit`` treats the X it is
    given as THIS rank's column shard of every block and all-reduces the per-iteration sums."""
    old = _state["comm"]
    set_comm(comm)
    try:
        yield comm
    finally:
        _state["comm"] = old


def cpu_quota():
    """CPUs this process may actually use: min(affinity, cgroup-v2/v1 CFS quota).  Containers often
    expose every host CPU while capping CPU time; BLAS pools sized from the visible CPU count then
    get throttled for whole scheduler periods (observed: 64 OpenBLAS threads on a 16-CPU quota stall
    the thread that drives the GPU by 40-2000 ms)."""
    import math
    import os
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            n = min(n, max(1, math.floor(int(q) / int(per))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


_tp_lock = __import__("threading").Lock()


def _thread_limits():
    """threadpoolctl's controller and the limit to apply (the library scan takes milliseconds: done once)"""
    with _tp_lock:
        if "tp" not in _state:
            try:
                import threadpoolctl
                tp = threadpoolctl.ThreadpoolController()
                # one process per GPU shares the node's quota with its siblings (torchrun exports LOCAL_WORLD_SIZE)
                import os
                siblings = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
                _state["tp_limit"] = max(1, min((cpu_quota() - 1) // siblings, 32))
                _state["tp"] = tp
            except Exception:  # pragma: no cover
                _state["tp"] = None


class _HostThreads:
    """Limit BLAS/OpenMP pools to the CPU quota for the host-side K x K / p x K algebra, restore on exit.  What
    ``threadpoolctl``'s ``limit()`` does, without its per-call inventory of every library (25 us at the head of every
    fit): the pools that are larger than the limit are set through their controllers and put back afterwards."""

    def __enter__(self):
        self.changed = []
        _thread_limits()
        tp = _state["tp"]
        if tp is None:
            return self
        limit = _state["tp_limit"]
        try:
            for lc in tp.lib_controllers:
                cur = lc.num_threads
                if cur is not None and cur > limit:
                    lc.set_num_threads(limit)
                    self.changed.append((lc, cur))
        except Exception:  # pragma: no cover -- an unusual pool: leave it alone
            pass
        return self

    def __exit__(self, *exc):
        for lc, cur in self.changed:
            try:
                lc.set_num_threads(cur)
            except Exception:  # pragma: no cover
                pass
        return False


def host_threads():
    return _HostThreads()
