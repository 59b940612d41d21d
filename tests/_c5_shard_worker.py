"""Worker of test_gpu_fullsize.py::test_c5_multiblock_nipals_full_size, part (b): BASELINE config[4]'s matrix
(n = 4096, four equal blocks, NIPALS, K = 10) column-sharded over TWO ranks that share the one GPU of the box.

Each rank generates its own column shard of every block in HBM (the generator is a function of the GLOBAL column
index, so the two shards together are bit for bit the matrix part (a) fitted on one GPU), fits it through the
product's sharded path (exchange over gloo, host-staged: RCCL refuses two ranks on one device) and checks
  * every replicated attribute bit-equal on both ranks (what SURVEY 8e's "same decisions everywhere" rests on),
  * the replicated attributes against part (a)'s (different summation order: 1e-8),
  * its slices of the sharded attributes (P_, W_, R_, beta_) against part (a)'s full arrays.
Inputs come from the directory C5_DIR written by the test (latent scores, Y, part (a)'s attributes)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import mbpls_amd  # noqa: E402
from mbpls_amd import MBPLS, DeviceMatrix  # noqa: E402
from mbpls_amd.comm import shard_columns  # noqa: E402
from tests import util  # noqa: E402

REPLICATED = ("Ts_", "U_", "V_", "A_", "A_corrected_", "explained_var_x_", "explained_var_y_", "explained_var_xblocks_")
TOL = 1e-8


def main():
    d = os.environ["C5_DIR"]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    mbpls_amd.set_device(0)
    ctx = mbpls_amd.get_context()
    comm = mbpls_amd.TorchDistComm(device="cuda:0")
    with np.load(os.path.join(d, "problem.npz")) as z:
        Tlat, Y, sizes, K = z["Tlat"], z["Y"], [int(s) for s in z["sizes"]], int(z["K"])
        ref = {k: z[k] for k in z.files if k in REPLICATED or k.startswith("T_")}
    n = Y.shape[0]
    goffs = np.concatenate(([0], np.cumsum(sizes)))
    ranges = shard_columns(sizes, rank, world)
    loffs = [0]
    for lo, hi in ranges:
        loffs.append(loffs[-1] + hi - lo)
    X = DeviceMatrix(ctx, n, loffs[-1], loffs)
    dT = ctx.from_host(Tlat)
    for b, (lo, hi) in enumerate(ranges):
        view = DeviceMatrix(ctx, n, hi - lo, None, storage=X.data_ptr + 8 * X.ld * loffs[b])
        view.fill_latent(dT, Tlat.shape[1], 0.5, 1234, int(goffs[b]) + lo)
        view.close()
    ctx.sync()
    dT.free()
    dist.barrier()
    try:
        with mbpls_amd.column_sharded(comm):
            mdl = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
            t0 = time.perf_counter()
            mdl = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
            wall = time.perf_counter() - t0
            assert mdl.nipals_iterations_ == [1] * K
            assert "never written" in mdl.fit_plan_["standardisation"], mdl.fit_plan_
            # ---- replicated attributes: bit-equal on both ranks
            mine = {k: np.asarray(getattr(mdl, k)) for k in REPLICATED}
            mine.update({f"T_{b}": np.asarray(t) for b, t in enumerate(mdl.T_)})
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            for other in everyone:
                for k, v in mine.items():
                    assert np.array_equal(v, other[k]), k
            # ---- ... and equal to the one-GPU fit of the whole matrix (same model, same data: no sign freedom)
            assert np.all(np.sum(mdl.Ts_ * ref["Ts_"], axis=0) > 0)
            for k, v in mine.items():
                e = util.relerr(v, ref[k])
                assert e < TOL, (k, e)
            # ---- this rank's slices of the sharded attributes against the full arrays of part (a)
            R = np.load(os.path.join(d, "R.npy"), mmap_mode="r")
            beta = np.load(os.path.join(d, "beta.npy"), mmap_mode="r")
            row = 0
            for b, (lo, hi) in enumerate(ranges):
                P = np.load(os.path.join(d, f"P{b}.npy"), mmap_mode="r")
                W = np.load(os.path.join(d, f"W{b}.npy"), mmap_mode="r")
                assert util.relerr(mdl.P_[b], P[lo:hi]) < TOL, ("P_", b)
                assert util.relerr(mdl.W_[b], W[lo:hi]) < TOL, ("W_", b)
                g = int(goffs[b])
                assert util.relerr(mdl.R_[row:row + hi - lo], R[g + lo:g + hi]) < TOL, ("R_", b)
                assert util.relerr(mdl.beta_[row:row + hi - lo], beta[g + lo:g + hi]) < TOL, ("beta_", b)
                row += hi - lo
            assert row == mdl.R_.shape[0] == loffs[-1]
            # ---- the sharded generic products (predict / transform all-reduce their n x K outputs)
            Ts_new = mdl.transform(X, Y)[0]
            assert util.relerr(Ts_new, mdl.Ts_) < TOL
            yhat = mdl.predict(X)
            assert util.relerr(yhat, mdl.y_scaler_.inverse_transform(mdl.Ts_.dot(mdl.V_.T))) < TOL
    finally:
        X.close()
    print(f"rank {rank}/{world}: C5 shard of {loffs[-1]} columns == the one-GPU fit, replicated attributes bit-equal; "
          f"fit {wall:.3f} s with both ranks on one GPU", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
