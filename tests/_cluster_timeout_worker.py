"""Worker of test_gpu_long.py::test_cluster_timeout_under_column_shards_is_a_collective_decision: two ranks share the GPU
(gloo), long-column cluster passes forced on a small problem, and ONE rank's hand-offs are made to give up (spin limit
of one poll).  The re-run on the two-phase passes issues collectives, so both ranks must take it together: the count of
time-outs rides on the all-reduce of the exchange buffer behind every host-driven pass and is summed once more when a
device queue returns (engine.check_cluster_passes) -- both ranks warn, both run again, both match the oracle."""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mbpls_amd  # noqa: E402
from mbpls_amd import MBPLS  # noqa: E402
from mbpls_amd.comm import shard_columns  # noqa: E402
from oracle import mbpls_oracle as orc  # noqa: E402
from tests import util  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    mbpls_amd.set_device(0)
    ctx = mbpls_amd.get_context()
    comm = mbpls_amd.TorchDistComm(device="cuda:0")
    n, sizes = 2100, [1300, 1500]               # (p > n: UNIPALS / KERNEL take their column-pass routes)
    cuts = np.cumsum(sizes)[:-1]
    ranges = shard_columns(sizes, rank, world)
    ctx.set_tunable("long_columns_from", 0)
    ctx.set_tunable("cluster_slab_rows", 128)
    ctx.set_tunable("cluster_exchange", 1)
    # (one response: a device-resident queue; several: host-driven sweeps; KERNEL: its maintained-products route.  SIMPLS
    # on long columns runs on the many-right-hand-side products, which have no hand-offs)
    os.environ["MBPLS_KERNEL_GRAM"] = "0"
    for method, q in (("NIPALS", 1), ("NIPALS", 2), ("UNIPALS", 1), ("UNIPALS", 2), ("KERNEL", 1)):
        X, Y = orc.make_latent(n, sum(sizes), q, r=6, seed=30 + q)
        full = list(np.split(X, cuts, axis=1))
        local = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]
        ref = orc.fit(full, Y, n_components=3, method=method, full_svd=True)
        ctx.set_tunable("cluster_passes", 1)
        ctx.__dict__.pop("_cluster_downgrade", None)
        assert ctx.cluster_columns(n)
        ctx.set_tunable("cluster_spin_limit", 1 if rank == 1 else 1 << 22)     # only rank 1's hand-offs give up
        with warnings.catch_warnings(record=True) as caught:
            warnings.simplefilter("always")
            with mbpls_amd.column_sharded(comm):
                mdl = MBPLS(n_components=3, method=method, full_svd=True).fit(local, Y)
        ctx.set_tunable("cluster_spin_limit", 1 << 22)
        told = [w for w in caught if "two-phase long-column passes" in str(w.message)]
        assert len(told) == 1, (method, q, rank, [str(w.message)[:80] for w in caught])   # on BOTH ranks
        assert ctx.cluster_passes == 0 and "two streaming phases" in mdl.fit_plan_["kernels"], mdl.fit_plan_
        assert "timed out" in mdl.fit_plan_["cluster_passes_off"], mdl.fit_plan_
        s = np.sign(np.sum(ref.Ts_ * mdl.Ts_, axis=0))
        assert util.relerr(mdl.Ts_ * s, ref.Ts_) < 1e-8, (method, q)
        for b, (lo, hi) in enumerate(ranges):
            assert util.relerr(mdl.P_[b] * s, ref.P_[b][lo:hi]) < 1e-8, (method, q, b)
        with mbpls_amd.column_sharded(comm):
            assert util.relerr(mdl.predict(local), orc.predict(ref, full)) < 1e-8, (method, q)
        assert ctx.cluster_status() == 0
    print(f"rank {rank}/{world}: a hand-off time-out on one rank re-ran the fit on both, == oracle", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
