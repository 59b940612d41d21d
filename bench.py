#!/usr/bin/env python3
"""bench.py -- MBPLS.fit() on device-resident synthetic wide matrices (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one complete ``MBPLS(n_components=10, method='NIPALS').fit(X, Y)`` on a matrix that is
already resident in HBM (feature-major): standardise + all components + fetching the fitted
attributes to the host.  X is generated on the device (planted latent structure) and is
NOT modified by fit (copy=True -> the standardise pass writes a working copy), so every step does
identical work.

N = 1 : BASELINE config[1]  single-block NIPALS n=4096 p=1,000,000 K=10 float64.
N > 1 : BASELINE config[4]  multiblock (4 blocks) NIPALS, n=4096, p = N x 1,000,000, feature columns of
        every block split evenly over the ranks (weak scaling: 32.8 GB of X per GPU), one RCCL
        all-reduce of [t~ (n x B), ||w~_b||^2] per NIPALS iteration via torch.distributed.

value  = reference-algorithm X traffic of the fit / wall time  [GB/s], whole job:
         units = sum_k (2 I_k + 3) + 1 (varx) + 3 (StandardScaler passes), 1 unit = 8 n p bytes
         (SURVEY 8d "algorithmic bytes per fit"; I_k = NIPALS iterations actually run).  The fused
         HIP path moves fewer bytes than that (reported as fused_units); the per-kernel roofline
         uses the kernel's own algorithmic bytes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); measured copy ceiling ~6290


def planted(n, q, r, seed, decay=0.85, noise_y=0.5):
    rng = np.random.RandomState(seed)
    T = rng.standard_normal((n, r))
    C = rng.standard_normal((r, q))
    Y = T.dot(C) + noise_y * rng.standard_normal((n, q))
    return T * decay ** np.arange(r), Y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--p", type=int, default=1_000_000, help="feature columns PER GPU")
    ap.add_argument("--K", type=int, default=10)
    ap.add_argument("--q", type=int, default=1)
    ap.add_argument("--method", default="NIPALS")
    ap.add_argument("--dist", action="store_true",
                    help="use the torch.distributed/RCCL exchange path and the N>1 block layout even at world size 1")
    ap.add_argument("--cpu-cols", type=int, default=65536, help="columns of the CPU-baseline sample (0 = skip)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix

    dist = None
    if world > 1 or a.dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # N processes share one node's CPU quota: keep torch's host pools from oversubscribing it
        os.environ.setdefault("OMP_NUM_THREADS", "1")
        import torch
        import torch.distributed as dist
        torch.set_num_threads(1)
        if os.environ.get("MBPLS_BENCH_SHARED_GPU") == "1":
            # dry run of the N>1 code path on a single-GPU box: all ranks drive cuda:0, exchange host-staged
            # over gloo (RCCL refuses duplicate devices).  Not a measurement configuration.
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local_rank}"))
        mbpls_amd.set_device(local_rank)
        mbpls_amd.set_comm(mbpls_amd.TorchDistComm(device=f"cuda:{local_rank}"))
    ctx = mbpls_amd.get_context()

    def barrier_sync():
        ctx.sync()
        if dist is not None:
            import torch
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()
        ctx.sync()

    n, p, K, q = a.n, a.p, a.K, a.q
    r = 32
    Ts, Y = planted(n, q, r, seed=0)
    # global matrix: world*p columns; N>1 = 4 equal blocks, every block split evenly over ranks
    B = 1 if dist is None else 4
    pb = p // B
    offs = [b * pb for b in range(B)] + [p]
    X = DeviceMatrix(ctx, n, p, offs)
    dT = ctx.from_host(Ts)
    if dist is None:
        X.fill_latent(dT, r, 0.5, 1234, 0)
    else:
        # rank r owns columns [r*pb_loc, (r+1)*pb_loc) of each global block (global block size world*pb)
        from mbpls_amd import _capi
        for b in range(B):
            view = DeviceMatrix(ctx, n, offs[b + 1] - offs[b], None, storage=X.data_ptr + 8 * X.ld * offs[b])
            view.fill_latent(dT, r, 0.5, 1234, (b * world + rank) * pb)
            view.close()
    ctx.sync()

    mdl = MBPLS(n_components=K, method=a.method, standardize=True, calc_all=True, copy=True)
    for _ in range(a.warmup):
        mdl.fit(X, Y)
    ctx.profile(True)
    ctx.profile_reset()
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        mdl.fit(X, Y)
    barrier_sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    prof = ctx.profile_get()
    ctx.profile(False)
    # Informational only, OUTSIDE the timed region and never part of `value`: the opt-in q=1 fixed-point mode
    # (MBPLS_NIPALS_Q1_FIXED_POINT=1) that skips the provably redundant second NIPALS pass per component.
    q1_opt = None
    if q == 1 and a.method == "NIPALS":
        os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = "1"
        try:
            mdl.fit(X, Y)
            barrier_sync()
            t1 = time.perf_counter()
            mdl.fit(X, Y)
            barrier_sync()
            q1_opt = {"fit_wall_s": round(time.perf_counter() - t1, 5), "nipals_iterations": list(mdl.nipals_iterations_),
                      "note": "opt-in mode, default off, not used for value/roofline"}
        finally:
            os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = "0"
            mdl.fit(X, Y)       # leave the estimator in its default state for the fields below

    iters = getattr(mdl, "nipals_iterations_", [2] * K)
    ref_units = sum(2 * i + 3 for i in iters) + 1 + 3
    # X traffic the HIP path really generates per fit = sum of the kernels' algorithmic bytes (HIP-event
    # profile of the timed region); with lazy deflation every pass after the standardisation is read-only
    fused_units = round(sum(v["bytes_per_launch"] * v["launches"] for v in prof.values()) / a.steps / (8.0 * n * p), 2)
    unit_bytes = 8.0 * n * p * world
    per_step = dt / a.steps
    value = ref_units * unit_bytes / per_step / 1e9

    out = None
    if rank == 0:
        dom = max(prof.items(), key=lambda kv: kv[1]["total_ms"]) if prof else (None, None)
        roof = None
        if dom[0]:
            k = dom[1]
            traffic = pmc_traffic(dom[0], n, p)
            roof = {"bound": "hbm", "kernel": dom[0], "achieved": round(k["GBps"], 1), "peak": HBM_PEAK_GBPS,
                    "unit": "GB/s", "frac": round(k["GBps"] / HBM_PEAK_GBPS, 4), "traffic": traffic,
                    "bytes_per_launch": k["bytes_per_launch"], "avg_ms": round(k["avg_ms"], 4),
                    "launches": k["launches"]}
        out = {
            "metric": "mbpls_fit_effective_hbm_GBps", "value": round(value, 1), "unit": "GB/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (planted latent structure, generated on device)",
            "config": {"workload": (f"single-block PLS NIPALS n={n} p={p} K={K} q={q} f64 (BASELINE config[1])"
                                    if dist is None else
                                    f"multiblock NIPALS n={n} total p={p * world} ({B} blocks) K={K} q={q} f64, "
                                    f"feature columns sharded over {world} GPUs (BASELINE config[4])"),
                       "method": a.method, "standardize": True, "calc_all": True, "nipals_iterations": iters},
            "fit_wall_s": round(per_step, 5), "fits_per_s": round(1.0 / per_step, 4),
            "ref_units_per_fit": ref_units, "fused_units_per_fit": fused_units,
            "fused_hbm_GBps": round(fused_units * unit_bytes / per_step / 1e9, 1),
            "roofline": roof,
            "kernels": {k: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4), "GBps": round(v["GBps"], 1)}
                        for k, v in prof.items()},
            "optional_q1_fixed_point": q1_opt,
        }
        if dist is None and a.cpu_cols > 0:
            out["cpu_baseline"] = cpu_baseline(ctx, X, Y, n, min(a.cpu_cols, p), K, a.method)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    X.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def pmc_traffic(kernel, n, p):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE in
    separate runs, gfx950 x2 read correction) -- only when they were taken on this exact shape, else None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_c2.json")) as f:
            d = json.load(f)
        if d["config"]["n"] == n and d["config"]["p"] == p:
            return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def cpu_baseline(ctx, X, Y, n, cols, K, method):
    """The oracle in faithful mode (= the reference's numpy op sequence, temporaries included) on a bounded
    column sample of the SAME device-generated matrix; also checks the GPU fit of that sample against it."""
    from mbpls_amd import MBPLS
    from oracle import mbpls_oracle as orc
    from mbpls_amd.runtime import cpu_quota
    Xs = X.download(0, cols)
    thr = cpu_quota()      # CPUs this container may really use (cgroup quota), not the visible count
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(limits=thr)
    except Exception:
        limiter = None
    t0 = time.perf_counter()
    ref = orc.fit(Xs, Y, n_components=K, method=method, standardize=True, calc_all=True, faithful=True)
    t_cpu = time.perf_counter() - t0
    if limiter is not None:
        limiter.restore_original_limits()
    iters = ref.get("nipals_iterations_", [2] * K)
    units = sum(2 * i + 3 for i in iters) + 1 + 3
    gbps = units * 8.0 * n * cols / t_cpu / 1e9
    mdl = MBPLS(n_components=K, method=method).fit(Xs, Y)
    s = np.sign(np.sum(ref.Ts_ * mdl.Ts_, axis=0))
    err = max(np.max(np.abs(mdl.Ts_ * s - ref.Ts_)) / np.max(np.abs(ref.Ts_)),
              np.max(np.abs(mdl.beta_ - ref.beta_)) / np.max(np.abs(ref.beta_)),
              np.max(np.abs(mdl.P_[0] * s - ref.P_[0])) / np.max(np.abs(ref.P_[0])))
    return {"value": round(gbps, 3), "unit": "GB/s", "cores": thr, "kind": "port",
            "sample": f"oracle faithful mode (reference numpy op sequence), first {cols} columns of the same matrix, "
                      f"n={n} K={K}, {t_cpu:.1f} s wall; BLAS threads = CPU quota = {thr} ({os.cpu_count()} logical CPUs visible)",
            "fit_wall_s": round(t_cpu, 3), "gpu_vs_oracle_relerr_on_sample": float(err)}


if __name__ == "__main__":
    main()
