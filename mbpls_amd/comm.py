"""Exchange step of the column-sharded fit.

Feature columns are split over ranks; every per-iteration quantity that is a sum over
features (un-normalised block scores t~, ||w~_b||^2, ||p_b||^2, X-sized sums of squares,
X W products, the Gram matrix) is all-reduced.  One process per GPU; the collective is
``torch.distributed`` (backend "nccl" == RCCL over xGMI on ROCm, "gloo" for CPU tests).
torch is plumbing here: it owns the exchange buffer and the communicator, nothing else.
"""
from __future__ import annotations

import numpy as np


class LocalComm:
    """Single device: no exchange."""
    rank = 0
    world_size = 1
    is_distributed = False

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
