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
    # all ranks must have taken identical decisions (iteration counts) and agree on the global scores
    print(f"rank {rank}: sharded == unsharded reference, worst rel err {worst:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


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
