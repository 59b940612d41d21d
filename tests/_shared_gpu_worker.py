"""Worker of test_gpu_rccl.py::test_two_ranks_share_one_gpu: world_size 2 with the REAL engine.

RCCL refuses two ranks on one device, and the GPU box has one GPU -- so the two ranks both drive cuda:0 and the
exchange goes through gloo (host-staged, mbpls_amd.comm.TorchDistComm).  Everything else is the product's
column-sharded path exactly as the multi-GPU run uses it: per-rank column shards of every block, the exchange
buffer filled by the HIP kernels, lazy-deflation sums, x_w / Gram / K x K reductions, identical loop-exit
decisions on both ranks."""
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
from oracle import mbpls_oracle_sparse as ors  # noqa: E402
from tests import util  # noqa: E402


def check(mdl, ref, ranges, full_new, local_new, Ynew, what, predict, transform):
    s = np.sign(np.sum(ref.Ts_ * mdl.Ts_, axis=0))
    assert util.relerr(mdl.Ts_ * s, ref.Ts_) < 1e-8, what
    assert util.relerr(mdl.U_ * s, ref.U_) < 1e-8, what
    for b, (lo, hi) in enumerate(ranges):
        assert util.relerr(mdl.P_[b] * s, ref.P_[b][lo:hi]) < 1e-8, what
        if isinstance(mdl.W_, list):
            assert util.relerr(mdl.W_[b] * s, ref.W_[b][lo:hi]) < 1e-8, what
            assert util.relerr(mdl.T_[b] * s, ref.T_[b]) < 1e-8, what
    offs = np.concatenate(([0], np.cumsum([len(p) for p in ref.P_])))
    rows = np.concatenate([np.arange(offs[b] + lo, offs[b] + hi) for b, (lo, hi) in enumerate(ranges)])
    assert util.relerr(mdl.beta_, ref.beta_[rows]) < 1e-8, what
    if "A_" in ref and np.size(ref.A_):
        assert util.relerr(mdl.A_, ref.A_) < 1e-8, what
        assert util.relerr(np.asarray(mdl.explained_var_xblocks_), np.asarray(ref.explained_var_xblocks_)) < 1e-8, what
    assert util.relerr(mdl.predict(local_new), predict(ref, full_new)) < 1e-8, what
    if mdl.method != "SIMPLS":
        Ts, T, U = mdl.transform(local_new, Ynew, return_block_scores=True)
        rTs, rT, rU = transform(ref, full_new, Ynew, return_block_scores=True)
        assert util.relerr(Ts * s, rTs) < 1e-8, what
        for b in range(len(ranges)):
            assert util.relerr(T[b] * s, rT[b]) < 1e-8, what


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    mbpls_amd.set_device(0)
    comm = mbpls_amd.TorchDistComm(device="cuda:0")
    n, q = 600, 2
    sizes = [1501, 2500, 2003]
    X, Y = orc.make_latent(n, sum(sizes), q, r=8, seed=5)
    Xn, Yn = orc.make_latent(50, sum(sizes), q, r=8, seed=6)
    cuts = np.cumsum(sizes)[:-1]
    ranges = shard_columns(sizes, rank, world)

    def shard(A):
        full = list(np.split(A, cuts, axis=1))
        return full, [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]

    full, local = shard(X)
    full_new, local_new = shard(Xn)
    with mbpls_amd.column_sharded(comm):
        for method, K in (("NIPALS", 6), ("UNIPALS", 4), ("KERNEL", 3), ("SIMPLS", 3)):
            ref = orc.fit(full, Y, n_components=K, method=method, full_svd=True)
            mdl = MBPLS(n_components=K, method=method).fit(local, Y)
            check(mdl, ref, ranges, full_new, local_new, Yn, method, orc.predict, orc.transform)
            if method == "NIPALS":
                # the exit test sits at 1e-14: a different summation order may move it by an iteration
                assert all(abs(a - b) <= 2 for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_))
        # one response: lazy deflation + implicit standardisation (X never written), sharded
        Y1, Yn1 = Y[:, :1], Yn[:, :1]
        for method, K, q1_route in (("NIPALS", 5, "1"), ("NIPALS", 5, "0"), ("UNIPALS", 3, "1")):
            os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = q1_route       # one pass per component (default) / the reference's two
            ref = orc.fit(full, Y1, n_components=K, method=method)
            mdl = MBPLS(n_components=K, method=method).fit(local, Y1)
            os.environ.pop("MBPLS_NIPALS_Q1_FIXED_POINT")
            check(mdl, ref, ranges, full_new, local_new, Yn1, method + " q=1", orc.predict, orc.transform)
            if method == "NIPALS":
                util.check_nipals_iterations(mdl, ref, K)
                assert ("fixed point" in mdl.fit_plan_["solver"]) == (q1_route == "1"), mdl.fit_plan_
        # explicit deflation and the streamed-column passes through the same exchange
        os.environ["MBPLS_DEFLATION"] = "explicit"
        ref = orc.fit(full, Y, n_components=3, method="NIPALS")
        check(MBPLS(n_components=3).fit(local, Y), ref, ranges, full_new, local_new, Yn, "explicit", orc.predict,
              orc.transform)
        del os.environ["MBPLS_DEFLATION"]
        mbpls_amd.get_context().set_tunable("long_columns_from", 0)
        check(MBPLS(n_components=3).fit(local, Y), ref, ranges, full_new, local_new, Yn, "long", orc.predict,
              orc.transform)
        for method in ("UNIPALS", "KERNEL", "SIMPLS"):         # ... and the other solvers' routes for long columns
            ref_m = orc.fit(full, Y, n_components=3, method=method, full_svd=True)
            check(MBPLS(n_components=3, method=method).fit(local, Y), ref_m, ranges, full_new, local_new, Yn,
                  "long " + method, orc.predict, orc.transform)
        mbpls_amd.get_context().set_tunable("long_columns_from", 8192)
        # missing values
        rng = np.random.RandomState(3)
        Xs = X.copy(); Xs[rng.rand(*X.shape) < 0.03] = np.nan
        Xns = Xn.copy(); Xns[rng.rand(*Xn.shape) < 0.03] = np.nan
        fulls, locals_ = shard(Xs)
        fulln, localn = shard(Xns)
        ref = ors.fit(fulls, Y, n_components=3)
        mdl = MBPLS(n_components=3, sparse_data=True).fit(locals_, Y)
        check(mdl, ref, ranges, fulln, localn, Yn, "sparse", ors.predict, ors.transform)
        # n >= p (mbpls.py:388-479, :580-692 and the NIPALS / SIMPLS paths on such data): row-sharded p x q operands
        sizes2 = [201, 300]
        X2, Y2 = orc.make_latent(700, sum(sizes2), q, r=8, seed=7)
        Xn2, Yn2 = orc.make_latent(40, sum(sizes2), q, r=8, seed=8)
        cuts2 = np.cumsum(sizes2)[:-1]
        ranges2 = shard_columns(sizes2, rank, world)

        def shard2(A):
            full = list(np.split(A, cuts2, axis=1))
            return full, [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges2)]

        full2, local2 = shard2(X2)
        full_new2, local_new2 = shard2(Xn2)
        for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
            ref = orc.fit(full2, Y2, n_components=3, method=method, full_svd=True)
            mdl = MBPLS(n_components=3, method=method).fit(local2, Y2)
            check(mdl, ref, ranges2, full_new2, local_new2, Yn2, method + " n>=p", orc.predict, orc.transform)
        # cross-validation under column shards: folds batched per pass over X (one response with two methods, three
        # responses with NIPALS) and the fold-by-fold loop, every rank against a loop of oracle fits on the full columns
        from mbpls_amd import cv_batch
        from mbpls_amd.model_selection import cross_val_predict
        nc, sizes_c = 48, [301, 400]
        Xc, Yc = orc.make_latent(nc, sum(sizes_c), 3, r=6, seed=9)
        cuts_c = np.cumsum(sizes_c)[:-1]
        full_c = list(np.split(Xc, cuts_c, axis=1))
        local_c = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full_c, shard_columns(sizes_c, rank, world))]
        calls, real = [], cv_batch.predict_folds
        cv_batch.predict_folds = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
        for method, qc, cv in (("NIPALS", 1, nc), ("KERNEL", 1, 6), ("NIPALS", 3, 8)):
            Yq = Yc[:, :qc]
            est = MBPLS(n_components=3, method=method)
            from sklearn.model_selection import KFold
            splits = list(KFold(cv).split(Xc))
            want = np.empty_like(Yq)
            for tr, te in splits:
                r = orc.fit([b[tr] for b in full_c], Yq[tr], n_components=3, method=method)
                want[te] = orc.predict(r, [b[te] for b in full_c])
            calls.clear()
            got = cross_val_predict(est, local_c, Yq, cv=cv)
            assert calls == [1], (method, qc, calls)                 # the batched path ran, on every rank
            loop = cross_val_predict(est, local_c, Yq, cv=cv, batch_folds=False)
            assert util.relerr(got, want) < 1e-8, (method, qc, util.relerr(got, want))
            assert util.relerr(loop, want) < 1e-8, (method, qc, util.relerr(loop, want))
        cv_batch.predict_folds = real
    print(f"rank {rank}/{world}: sharded fit on a shared GPU == oracle", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
