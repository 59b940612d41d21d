"""Process-wide device context and (optional) column-sharding communicator."""
from __future__ import annotations

import contextlib

from . import _capi
from .comm import LocalComm

_contexts = {}
_state = {"device": 0, "comm": LocalComm()}


def set_device(device):
    _state["device"] = int(device)


def get_context(device=None):
    """The shared ``Context`` of a device (created on first use; raises if no GPU / no library)."""
    d = _state["device"] if device is None else int(device)
    ctx = _contexts.get(d)
    if ctx is None or ctx.handle is None:
        ctx = _capi.Context(d)
        _contexts[d] = ctx
    return ctx


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
