"""Worker of test_gpu_rccl.py::test_ranks_agree_bit_for_bit: world_size >= 2 with the REAL engine.

What SURVEY 8e relies on -- "all ranks receive bit-identical sums, so every rank takes the same loop-exit decision" --
is asserted directly: every REPLICATED fitted attribute (Ts_, T_, U_, V_, A_, A_corrected_, the explained variances,
nipals_iterations_, nipals_converged_) is gathered from all ranks and must be bit-equal; the SHARDED ones (P_, W_, R_,
beta_) are put together from the ranks' column ranges and, like the replicated ones, compared with the CPU oracle.
MBPLS_TEST_SHARED_GPU=1: both ranks drive cuda:0 over gloo (the single-GPU pool); otherwise one rank per GPU and the C
ABI's RCCL communicator (BASELINE config[4]'s exchange)."""
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

REPLICATED = ("Ts_", "U_", "V_", "A_", "A_corrected_", "explained_var_x_", "explained_var_y_", "explained_var_xblocks_")


def main():
    shared = os.environ.get("MBPLS_TEST_SHARED_GPU") == "1"
    local_rank = 0 if shared else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if shared:
        dist.init_process_group("gloo")
    else:
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world >= 2
    mbpls_amd.set_device(local_rank)
    if shared:
        comm = mbpls_amd.TorchDistComm(device="cuda:0")
    else:
        comm = mbpls_amd.RcclComm.from_torch(mbpls_amd.get_context())
        assert comm.info()["rccl_ranks"] == world
    n = 700
    sizes = [2100, 1300, 2601, 999]                      # four ragged blocks (BASELINE config[4] has four)
    cuts = np.cumsum(sizes)[:-1]
    ranges = shard_columns(sizes, rank, world)
    offs = np.concatenate(([0], np.cumsum(sizes)))
    # (one response twice: the default one-pass-per-component route, then the reference's two passes per component,
    # whose stop decision is the one that must not differ between ranks)
    for q, K, q1_route in ((1, 6, "1"), (1, 6, "0"), (3, 3, "1")):
        os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = q1_route
        X, Y = orc.make_latent(n, sum(sizes), q, r=8, seed=11 + q)
        full = list(np.split(X, cuts, axis=1))
        local = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]
        ref = orc.fit(full, Y, n_components=K, method="NIPALS")
        with mbpls_amd.column_sharded(comm):
            mdl = MBPLS(n_components=K, method="NIPALS").fit(local, Y)
        mine = {k: np.asarray(getattr(mdl, k)) for k in REPLICATED}
        mine.update({f"T_{b}": t for b, t in enumerate(mdl.T_)})
        mine["nipals_iterations_"] = np.asarray(mdl.nipals_iterations_)
        mine["nipals_converged_"] = np.asarray(getattr(mdl, "nipals_converged_", []))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        for other in everyone:                            # bit-equal on every rank
            for k, v in mine.items():
                assert np.array_equal(v, other[k]), (k, q)
        if q == 1:
            util.check_nipals_iterations(mdl, ref, K)
            assert ("fixed point" in mdl.fit_plan_["solver"]) == (q1_route == "1"), mdl.fit_plan_
        else:
            assert all(abs(a - b) <= 2 for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_))
        # the sharded attributes, assembled over the ranks, against the oracle
        parts = [None] * world
        dist.all_gather_object(parts, dict(P=[np.asarray(p) for p in mdl.P_], W=[np.asarray(w) for w in mdl.W_],
                                           R=np.asarray(mdl.R_), beta=np.asarray(mdl.beta_)))
        s = np.sign(np.sum(ref.Ts_ * mdl.Ts_, axis=0))
        assert util.relerr(mdl.Ts_ * s, ref.Ts_) < 1e-8 and util.relerr(mdl.U_ * s, ref.U_) < 1e-8
        assert util.relerr(mdl.A_, ref.A_) < 1e-8
        Rfull, bfull = np.zeros_like(ref.R_), np.zeros_like(ref.beta_)
        for r_, part in enumerate(parts):
            rr = shard_columns(sizes, r_, world)
            row = 0
            for b, (lo, hi) in enumerate(rr):
                assert util.relerr(part["P"][b] * s, ref.P_[b][lo:hi]) < 1e-8, (q, b)
                assert util.relerr(part["W"][b] * s, ref.W_[b][lo:hi]) < 1e-8, (q, b)
                Rfull[offs[b] + lo:offs[b] + hi] = part["R"][row:row + hi - lo]
                bfull[offs[b] + lo:offs[b] + hi] = part["beta"][row:row + hi - lo]
                row += hi - lo
        assert util.relerr(Rfull * s, ref.R_) < 1e-8 and util.relerr(bfull, ref.beta_) < 1e-8
    print(f"rank {rank}/{world}: replicated attributes bit-equal on all ranks, all == oracle", flush=True)
    dist.barrier()
    if hasattr(comm, "close"):
        comm.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
