"""Worker of test_gpu_rccl.py: the product's sharded path with the REAL engine and RCCL (backend nccl),
run under torch.distributed.run.  With one visible GPU world_size is 1: the all-reduce is then a
self-reduction, but every piece of plumbing (torch-owned exchange buffer filled by the HIP kernels on
the null stream, RCCL collective on it, device->host) is the one the 8-GPU run uses."""
import os
import sys

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
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    rank, world = dist.get_rank(), dist.get_world_size()
    mbpls_amd.set_device(local_rank)
    which = os.environ.get("MBPLS_TEST_COMM", "torch")
    if which == "rccl":
        # the C ABI's own communicator: ncclAllReduce on the context's stream, bootstrapped through torch's store
        comm = mbpls_amd.RcclComm.from_torch(mbpls_amd.get_context())
        info = comm.info()
        assert info["world_size"] == world and info["rank"] == rank and info["rccl_version"] > 0, info
    else:
        comm = mbpls_amd.TorchDistComm(device=f"cuda:{local_rank}")
    q = int(os.environ.get("MBPLS_TEST_Q", "2"))
    X, Y = orc.make_latent(512, 6000, q, r=8, seed=5)
    sizes = [1500, 2500, 2000]
    full = list(np.split(X, np.cumsum(sizes)[:-1], axis=1))
    ranges = shard_columns(sizes, rank, world)
    local = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]
    # (one-response NIPALS: the default one-pass-per-component route, then the reference's two passes per component)
    for method, q1_route in (("NIPALS", "1"), ("NIPALS", "0"), ("UNIPALS", "1"), ("KERNEL", "1"), ("SIMPLS", "1")):
        if q1_route == "0" and q != 1:
            continue
        os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = q1_route
        ref = orc.fit(full, Y, n_components=3, method=method, full_svd=True)
        with mbpls_amd.column_sharded(comm):
            mdl = MBPLS(n_components=3, method=method).fit(local, Y)
            yh = mdl.predict(local)
        os.environ.pop("MBPLS_NIPALS_Q1_FIXED_POINT")
        s = np.sign(np.sum(ref.Ts_ * mdl.Ts_, axis=0))
        assert util.relerr(mdl.Ts_ * s, ref.Ts_) < 1e-8, method
        for b, (lo, hi) in enumerate(ranges):
            assert util.relerr(mdl.P_[b] * s, ref.P_[b][lo:hi]) < 1e-8, method
        assert util.relerr(yh, orc.predict(ref, full)) < 1e-8, method
        if method == "NIPALS" and q == 1:
            util.check_nipals_iterations(mdl, ref)
            assert ("fixed point" in mdl.fit_plan_["solver"]) == (q1_route == "1"), mdl.fit_plan_
    # n >= p (mbpls.py:388-479, :580-692): the leading singular vector of a p x q matrix whose rows are split over the
    # ranks, (X^T X) w from every rank's columns
    X, Y = orc.make_latent(700, 500, q, r=8, seed=6)
    sizes = [200, 300]
    full = list(np.split(X, np.cumsum(sizes)[:-1], axis=1))
    ranges = shard_columns(sizes, rank, world)
    local = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]
    for method in ("UNIPALS", "KERNEL"):
        ref = orc.fit(full, Y, n_components=3, method=method, full_svd=True)
        with mbpls_amd.column_sharded(comm):
            mdl = MBPLS(n_components=3, method=method).fit(local, Y)
            yh = mdl.predict(local)
        s = np.sign(np.sum(ref.Ts_ * mdl.Ts_, axis=0))
        assert util.relerr(mdl.Ts_ * s, ref.Ts_) < 1e-8, ("n >= p", method)
        for b, (lo, hi) in enumerate(ranges):
            assert util.relerr(mdl.P_[b] * s, ref.P_[b][lo:hi]) < 1e-8, ("n >= p", method)
        assert util.relerr(yh, orc.predict(ref, full)) < 1e-8, ("n >= p", method)
    print(f"rank {rank}/{world}: RCCL-sharded fit == oracle ({which}, q={q})", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
