"""Worker of test_sharded_gloo.py: run under torch.distributed.run with WORLD_SIZE=2 (gloo, CPU).

Each rank takes its column shard of every block (mbpls_amd.comm.shard_columns), runs the SAME host
drivers as the product (mbpls_amd/solvers.py) with the test-only numpy engine, all-reducing through
TorchDistComm, and checks the result against the un-sharded reference fixture."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch.distributed as dist  # noqa: E402

from mbpls_amd import solvers  # noqa: E402
from mbpls_amd.comm import TorchDistComm, shard_columns  # noqa: E402
from mbpls_amd.scaler import ColumnScaler  # noqa: E402
from tests import util  # noqa: E402
from tests.numpy_engine import NumpyEngine  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = TorchDistComm(device=None)
    worst = 0.0
    # p > n and n >= p: the n >= p branches of UNIPALS / KERNEL (mbpls.py:388-479, :580-692) take their leading singular
    # vector from a p x q matrix whose ROWS are split over the ranks, and X^T X w needs every rank's columns
    for fixture in ("matrix_pgtn", "matrix_ngtp"):
        worst = max(worst, one_fixture(util.load(fixture), rank, world, comm))
    worst = max(worst, random_sweep(rank, world, comm))
    worst = max(worst, random_sweep_missing(rank, world, comm))
    # all ranks must have taken identical decisions (iteration counts) and agree on the global scores
    print(f"rank {rank}: sharded == unsharded reference, worst rel err {worst:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


def random_sweep(rank, world, comm, cases=30):
    """random shapes / block splits / methods / routes, every problem column-sharded over the ranks: each rank's result
    against the oracle's attributes as that rank holds them (tests/fuzz._local_view)"""
    from oracle import mbpls_oracle as orc
    from tests import fuzz
    rng = np.random.RandomState(21)                       # (the same problems on every rank)
    worst = 0.0
    for c in range(cases):
        n = int(rng.choice([rng.randint(8, 40), rng.randint(40, 160)]))
        nb = int(rng.randint(1, 4))
        sizes = [int(rng.randint(2 * world, 2 * world + 60)) for _ in range(nb)]
        p = sum(sizes)
        q = int(rng.choice([1, 1, 2, 3]))
        method = str(rng.choice(["NIPALS", "NIPALS", "UNIPALS", "KERNEL", "SIMPLS"]))
        K = int(min(rng.randint(1, 5), max(1, min(n // 4, p))))
        std, ca = bool(rng.rand() < 0.7), bool(rng.rand() < 0.8)
        os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = "0" if rng.rand() < 0.5 else "1"
        X, Y = orc.make_latent(n, p, q, r=4, seed=4000 + c)
        X = X + rng.uniform(-3, 3, size=p) * (rng.rand() < 0.5)
        offs = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
        full = [X[:, offs[b]:offs[b + 1]] for b in range(nb)]
        ref = orc.fit(full, Y, n_components=K, method=method, standardize=std, full_svd=True, calc_all=ca, max_iter=3000)
        if method == "NIPALS" and max(ref.nipals_iterations_) >= 3000:
            continue
        ranges = shard_columns(sizes, rank, world)
        local = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]
        Ys = ColumnScaler().fit(Y).transform(Y) if std else Y
        eng = NumpyEngine(comm).load_host_blocks(local)
        if method == "NIPALS":
            res = solvers.fit_nipals(eng, Ys, K, std, calc_all=ca)
        elif method == "UNIPALS":
            res = solvers.fit_unipals(eng, Ys, K, std)
        elif method == "KERNEL":
            res = solvers.fit_kernel(eng, Ys, K, std, ca)
        else:
            res = solvers.fit_simpls(eng, Ys, K, std)
        what = f"rank{rank} case {c}: n={n} sizes={sizes} q={q} K={K} {method} std={std} ca={ca}"
        worst = max(worst, util.compare(util.model_to_flat(res), fuzz._local_view(ref, ranges, offs), 1e-6, what=what))
    os.environ.pop("MBPLS_NIPALS_Q1_FIXED_POINT", None)
    assert worst < 1e-7, worst
    return worst


def random_sweep_missing(rank, world, comm, cases=14):
    """the same with missing values (sparse_data=True): the masked sums A, D and the weight norms are what the ranks
    exchange; one response on both routes, several responses"""
    import warnings
    from oracle import mbpls_oracle as orc
    from oracle import mbpls_oracle_sparse as ors
    from tests import fuzz
    rng = np.random.RandomState(33)
    worst = 0.0
    for c in range(cases):
        n = int(rng.randint(20, 100))
        nb = int(rng.randint(1, 4))
        sizes = [int(rng.randint(2 * world + 6, 2 * world + 60)) for _ in range(nb)]
        p = sum(sizes)
        q = int(rng.choice([1, 1, 2]))
        K = int(min(rng.randint(1, 4), n // 4))
        std = bool(rng.rand() < 0.7)
        os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = "0" if rng.rand() < 0.5 else "1"
        X, Y = orc.make_latent(n, p, q, r=4, seed=8000 + c)
        X[rng.rand(n, p) < 0.04] = np.nan
        offs = np.concatenate(([0], np.cumsum(sizes))).astype(np.int64)
        full = [X[:, offs[b]:offs[b + 1]] for b in range(nb)]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = ors.fit(full, Y, n_components=K, standardize=std, max_iter=3000)
        if max(ref.nipals_iterations_) >= 3000:
            continue
        ranges = shard_columns(sizes, rank, world)
        local = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]
        Ys = ColumnScaler().fit(Y).transform(Y) if std else Y
        eng = NumpyEngine(comm).load_host_blocks(local)
        eng.sparse = True
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            res = solvers.fit_nipals(eng, Ys, K, std, calc_all=True)
        what = f"rank{rank} missing values, case {c}: n={n} sizes={sizes} q={q} K={K} std={std}"
        worst = max(worst, util.compare(util.model_to_flat(res), fuzz._local_view(ref, ranges, offs), 1e-6, what=what))
    os.environ.pop("MBPLS_NIPALS_Q1_FIXED_POINT", None)
    assert worst < 1e-7, worst
    return worst


def one_fixture(g, rank, world, comm):
    splits = [int(s) for s in g["splits"]]
    cuts = np.cumsum(splits)[:-1]
    worst = 0.0
    for q, std, ca in ((1, 1, 1), (3, 1, 1), (1, 0, 0), (3, 0, 1), (3, 1, 0)):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        Ys = ColumnScaler().fit(Y).transform(Y) if std else Y
        for blocks in ("single", "multi"):
            full = [X] if blocks == "single" else list(np.split(X, cuts, axis=1))
            ranges = shard_columns([b.shape[1] for b in full], rank, world)
            local = [np.ascontiguousarray(b[:, lo:hi]) for b, (lo, hi) in zip(full, ranges)]
            for m in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
                eng = NumpyEngine(comm).load_host_blocks(local)
                if m == "NIPALS":
                    res = solvers.fit_nipals(eng, Ys, 3, bool(std), calc_all=bool(ca))
                elif m == "UNIPALS":
                    res = solvers.fit_unipals(eng, Ys, 3, bool(std))
                elif m == "KERNEL":
                    res = solvers.fit_kernel(eng, Ys, 3, bool(std), bool(ca))
                else:
                    res = solvers.fit_simpls(eng, Ys, 3, bool(std))
                ref = util.sub(g, f"ref/q{q}/{blocks}/{m}/std{std}/ca{ca}")
                # feature-indexed attributes: compare this rank's rows with the same rows of the reference
                flat = util.model_to_flat(res)
                sref = {}
                for k, v in ref.items():
                    base = k.split("/")[0]
                    if base in ("predict", "Ts_new", "T_new", "U_new", "x_mean", "x_scale", "y_mean", "y_scale"):
                        continue
                    if v.size == 0:
                        continue
                    if base in ("P_", "W_", "W_non_normal_") and "/" in k:
                        lo, hi = ranges[int(k.split("/")[1])]
                        sref[k] = v[lo:hi]
                    elif base in ("R_", "beta_", "W_concat_") or (base == "W_" and m == "SIMPLS"):
                        rows = np.concatenate([np.arange(o + lo, o + hi) for o, (lo, hi) in
                                               zip(np.concatenate(([0], np.cumsum([b.shape[1] for b in full])[:-1])), ranges)])
                        sref[k] = v[rows]
                    else:
                        sref[k] = v
                worst = max(worst, util.compare(flat, sref, 1e-9, what=f"rank{rank}/q{q}/{blocks}/{m}/std{std}/ca{ca}"))
    return worst


if __name__ == "__main__":
    main()
