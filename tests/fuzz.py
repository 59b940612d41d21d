"""Randomised parity sweep: random shapes / block splits / methods / settings (incl. column offsets and missing
values), the HIP path against the oracles.  Used by test_gpu_fuzz.py and tools/fuzz_parity.py."""
import warnings

import numpy as np

from oracle import mbpls_oracle as orc
from oracle import mbpls_oracle_sparse as ors
from tests import util


def _local_view(ref, ranges, offs):
    """the oracle's attributes as ONE RANK of a column-sharded fit holds them: n-side attributes whole, p-side attributes
    restricted to the rank's columns of every block"""
    flat = util.model_to_flat(ref)
    rows = np.concatenate([np.arange(offs[b] + lo, offs[b] + hi) for b, (lo, hi) in enumerate(ranges)])
    out = {}
    for k, v in flat.items():
        base = k.split("/")[0]
        if base in ("W_", "P_", "W_non_normal_") and "/" in k:
            lo, hi = ranges[int(k.split("/")[1])]
            out[k] = np.asarray(v)[lo:hi]
        elif base in ("R_", "beta_", "W_concat_") or (base == "W_" and "/" not in k):
            out[k] = np.asarray(v)[rows] if np.asarray(v).shape[0] == offs[-1] else v
        else:
            out[k] = v
    return out


def run(cases, seed, verbose=True, big=False, long=False, only=None, raise_errors=False, shard=None, wide=False):
    """``big``: register-resident columns (1024 < n <= 8192); ``long``: long columns (8192 < n <= 40000: clusters of
    workgroups / two-phase passes, the multi-workgroup step from 32768 on) on narrow matrices.  One-response NIPALS takes
    the default route (one pass per component) or the reference's two passes, at random."""
    import os
    from mbpls_amd import MBPLS
    from mbpls_amd.plan import FitOptions
    warnings.filterwarnings("ignore")
    try:        # the oracle's BLAS pool at the CPU quota of the box (see tests/conftest.py), for runs outside pytest too
        import threadpoolctl
        from mbpls_amd import runtime
        _cap = threadpoolctl.threadpool_limits(limits=max(1, min(runtime.cpu_quota(), 32)))    # noqa: F841
    except Exception:
        pass
    rng = np.random.RandomState(seed)
    print_ = print if verbose else (lambda *a, **k: None)
    worst, bad = 0.0, 0
    reused = None                      # one estimator object refitted with new parameters now and then (stale state)
    for c in range(cases):
        n = int(rng.choice([rng.randint(3, 40), rng.randint(40, 700), rng.choice([64, 128, 256, 512, 1024])]))
        if big:      # register-resident columns (1024 < n <= 8192), ragged and odd sample counts included
            n = int(rng.choice([rng.randint(1025, 4096), rng.randint(4097, 8192), rng.choice([2048, 4096, 4097, 8191, 8192])]))
        if long:
            n = int(rng.choice([rng.randint(8193, 20000), rng.randint(20000, 40000), rng.randint(40000, 66000),
                                rng.choice([8194, 16384, 32768, 32769, 65536, 65537, 131072])]))
        p = int(rng.choice([rng.randint(1, 30), rng.randint(30, 3000)]))
        if long:
            p = int(rng.choice([rng.randint(1, 30), rng.randint(30, 400)]))
        if wide:                               # few samples x very many features: the everyday multiblock-PLS shape
            n = int(rng.choice([rng.randint(3, 40), rng.randint(40, 300), rng.choice([16, 64, 128, 256])]))
            p = int(rng.choice([rng.randint(3000, 30000), rng.randint(30000, 200000)]))
            if os.environ.get("MBPLS_FUZZ_WIDE_P"):     # probes beyond the sweep: "lo,hi[,nmax]", e.g. columns above the
                lo, hi, *nmax = (int(v) for v in os.environ["MBPLS_FUZZ_WIDE_P"].split(","))   # copier's bounce size
                p = int(rng.randint(lo, hi))
                n = min(n, nmax[0]) if nmax else n
        if shard is not None:
            p = max(p, 2 * shard[2])           # (a rank without a single column is not a configuration anyone runs)
        q = int(rng.choice([1, 1, 2, 3, 5])) if big else int(rng.choice([1, 1, 2, 3, int(rng.randint(4, 13))]))
        if not (big or long or wide) and rng.rand() < 0.08:
            q = int(rng.randint(30, 48))       # more responses than a 32 x 32 tile of the small device products holds
        kmax = int(rng.choice([rng.randint(1, 5), rng.randint(5, 10)]))
        if big or long or wide:
            kmax = min(kmax, 4)                # (the CPU oracle iterates hundreds of times per component for q > 1)
        K = int(min(kmax, max(1, min(n - 1, p))))
        method = str(rng.choice(["NIPALS", "NIPALS", "UNIPALS", "KERNEL", "SIMPLS"]))
        std = bool(rng.rand() < 0.7)
        nb = int(min(p, rng.randint(1, 4)))
        if shard is not None:
            nb = int(min(nb, max(1, p // (2 * shard[2]))))     # every rank holds at least two columns of every block
        cuts = np.sort(rng.choice(np.arange(1, p), nb - 1, replace=False)) if nb > 1 else np.array([], dtype=int)
        if shard is not None and nb > 1:                       # (blocks of at least 2 x world columns)
            w2 = 2 * shard[2]
            cuts = np.sort(rng.choice(np.arange(1, p // w2), nb - 1, replace=False)) * w2 if p // w2 > nb else np.array([], dtype=int)
            nb = len(cuts) + 1
        X, Y = orc.make_latent(n, p, q, r=min(4, p), seed=1000 + c)
        # column offsets (implicit standardisation): mostly modest, now and then far beyond the eight standard deviations
        # from which the raw-space passes centre in registers -- raw spectra (a device queue that is stopped and queued
        # again, or starts with centring passes because the last fit of the shape needed them: engine._offset_hint)
        offset_scale = float(rng.choice([1.0, 1.0, 1.0, 10.0, 300.0]))
        X = X + rng.uniform(-3, 3, size=p) * (rng.rand() < 0.5) * offset_scale
        sp = (lambda A: A) if nb == 1 else (lambda A: [np.ascontiguousarray(b) for b in np.split(A, cuts, axis=1)])
        sparse = method == "NIPALS" and n >= 20 and p >= 20 and rng.rand() < 0.4
        if sparse:
            X[rng.rand(n, p) < rng.choice([0.01, 0.05])] = np.nan
        two_pass = method == "NIPALS" and q == 1 and not sparse and rng.rand() < 0.5
        calc_all = bool(rng.rand() < 0.8)
        norm = [2, 2, 1, np.inf, "fro"][int(rng.randint(0, 5))] if (method == "NIPALS" and not sparse) else 2
        if big or long or wide:
            norm = 2
        nn = int(rng.randint(1, 12))                                       # new samples for predict / transform
        Xn, Yn = orc.make_latent(nn, p, q, r=min(4, p), seed=5000 + c)
        if sparse:
            Xn[rng.rand(nn, p) < 0.05] = np.nan
        # (every random decision of a case is drawn HERE, whatever happens to its fit: ``only`` replays the same case)
        form = int(rng.randint(0, 8))
        do_cv, cv_pick = bool(rng.rand() < 0.2), int(rng.randint(0, 3))
        reuse = bool(rng.rand() < 0.3)
        # a non-default route now and then (the switches of mbpls_amd/plan.py): same numbers on every route
        # (lower-case keys: fallback forms of mbpls_amd.plan.FALLBACK_FORMS, forced through FitOptions.forced)
        routes = [{}, {}, {}, {"MBPLS_DEFLATION": "explicit"}, {"MBPLS_IMPLICIT_STD": "0"}, {"MBPLS_RESIDENT": "0"},
                  {"nipals_subspace": False}, {"nipals_subspace": False, "nipals_gram": False},
                  {"unipals_nodeflate": False}, {"r_incremental": False}, {"stream_stats": False},
                  {"MBPLS_KERNEL_IMPLICIT": "1"}, {"MBPLS_KERNEL_IMPLICIT": "0"}, {"nipals_qspace": False},
                  {"MBPLS_KERNEL_GRAM": "1"}, {"MBPLS_KERNEL_GRAM": "0"},
                  {"MBPLS_KERNEL_GRAM": "1", "MBPLS_KERNEL_IMPLICIT": "1"}]
        route = routes[int(rng.randint(0, len(routes)))]
        # missing values: every third input form through the streamed-column kernel that n > 65536 takes (any n runs on it)
        stream_nan = sparse and form % 3 == 0
        nan_from = 0 if stream_nan else (65536 if (sparse and form % 3 == 1) else None)   # ... and the register shapes to 65536
        # launch tunables of the C ABI (same numbers whatever they are): missing-value kernel choice; long columns on
        # two-phase passes instead of clusters, on clusters of many small slabs, or (n <= 65536) in spilling registers
        tun = {}
        if nan_from is not None:
            tun["nan_long_from"] = (nan_from, 8192)
            if stream_nan and form % 2:       # the one-workgroup-per-chunk streamed kernel instead of the two phases
                tun["nan_long_two_phase"] = (0, 1)
        if long and not sparse:
            pick = (form + c) % 5
            if pick == 1:
                tun["cluster_passes"] = (0, 1)
            elif pick == 2:
                tun["cluster_slab_rows"] = (1024, 8192)
            elif pick == 3 and n <= 65536:
                tun["long_columns_from"] = (65536, 8192)
        tag = (f"case {c}: n={n} p={p} q={q} K={K} {method} std={std} blocks={nb} sparse={sparse} two_pass={two_pass} "
               f"calc_all={calc_all} norm={norm} form={form}" + (f" route={route}" if route else "")
               + (f" tunables={ {k: v[0] for k, v in tun.items()} }" if tun else ""))
        if only is not None and c != only:                 # (same random sequence, one case fitted: reproducing a failure)
            continue
        if os.environ.get("MBPLS_FUZZ_TRACE"):             # every case announced before its fit (which rank stopped where)
            print(f"[trace pid {os.getpid()}] {tag}", flush=True)
        try:
            if sparse:
                ref = ors.fit(sp(X), Y, n_components=K, standardize=std, max_iter=3000)
            else:
                ref = orc.fit(sp(X), Y, n_components=K, method=method, standardize=std, full_svd=True, calc_all=calc_all,
                              nipals_convergence_norm=norm, max_iter=3000)
        except Exception as e:
            print_(tag, "oracle failed:", type(e).__name__); continue
        # SIMPLS: the reference's Gram-Schmidt recurrence (mbpls.py:1015-1018) is classical and loses orthogonality
        # itself -- |Ts^T Ts - I| from 1e-8 at eight or nine components up to 1e-2 once K approaches the number of
        # samples, on uncentred data most of all.  A fit can follow it to about that defect, not to 1e-6: where 50 x
        # the ORACLE's own defect exceeds 1e-8 that is the bound of the case (asserted; the case then stays out of
        # ``worst``), everywhere else the usual 1e-6 (tests/test_gpu_parity.py::test_simpls_follows_the_reference_...)
        # The defect is not the whole of it: with K next to the rank of the (centred) data the last score u - T (T^T u)
        # (:1017-1019) is what cancellation leaves of a vector that lies in span(T) to six or seven digits.  Seed 1707,
        # case 133 (n = 9, K = 8, standardised: rank 8): Ts orthonormal to 2e-10, yet ONE ULP on random elements of X
        # moves the oracle's own U_[:, 7] by 7e-7 and V_[:, 7] by 2e-5.  The oracle refitted on inputs perturbed by one
        # ulp measures that; the bound of a SIMPLS case is 50 x the larger of the two figures.
        # The same measure bounds fits of UNCENTRED data whose columns sit tens to hundreds of standard deviations from
        # zero (standardize=False on raw spectra): X X^T is then one huge rank-one term plus the structure, and the
        # reference's own later components move by 1e-7 under one ulp (seed 5707, case 198: KERNEL, offsets to 1400
        # sigma, Ts orthonormal to 5e-9 only, W_concat_ of the fit 2.5e-6 off).
        bound = None
        if method == "SIMPLS" or (not std and offset_scale >= 10.0 and not sparse):
            defect = float(np.max(np.abs(ref.Ts_.T.dot(ref.Ts_) - np.eye(K))))
            prng = np.random.RandomState(77 + c)
            Xp = X * (1.0 + 2.220446049250313e-16 * prng.choice([-1.0, 0.0, 1.0], size=X.shape))
            try:
                ref_p = orc.fit(sp(Xp), Y, n_components=K, method=method, standardize=std, full_svd=True,
                                calc_all=calc_all, nipals_convergence_norm=norm, max_iter=3000)
                sens = float(util.compare(util.model_to_flat(ref_p), util.model_to_flat(ref), np.inf, what=tag))
            except Exception:
                sens = 0.0
            defect = max(defect, sens)
            if 50.0 * defect > 1e-8:
                bound = 50.0 * defect
        tol_case = max(1e-6, bound or 0.0)
        if method == "NIPALS" and max(ref.nipals_iterations_) >= 3000:
            # the REFERENCE's loop does not converge here (masked alternating regressions of some missing-value patterns,
            # structure-free deep components: it would run forever, :841): nothing to compare with
            print_(tag, "skipped: the reference's NIPALS loop does not converge (iteration cap of the oracle)"); continue
        try:
            os.environ["MBPLS_NIPALS_Q1_FIXED_POINT"] = "0" if two_pass else "1"
            os.environ.update({k: v for k, v in route.items() if k.isupper()})
            FitOptions.forced.clear()
            FitOptions.forced.update({k: v for k, v in route.items() if not k.isupper()})
            if tun:
                import mbpls_amd
                for key, (val, _) in tun.items():
                    mbpls_amd.get_context().set_tunable(key, val)
            params = dict(n_components=K, method=method, standardize=std, full_svd=True, sparse_data=sparse,
                          calc_all=(calc_all or sparse), nipals_convergence_norm=norm)
            if reuse and reused is not None:
                est = reused.set_params(**params)          # the SAME object, fitted before with other data and settings
            else:
                est = MBPLS(**params)
            reused = est
            Yin = Y[:, 0] if (q == 1 and form % 2) else Y                  # 1-D responses now and then
            if shard is not None:
                # column shards: this rank fits ITS columns of every block; compared with the oracle's attributes as the
                # rank holds them (n-side whole, p-side sliced); predictions and scores of new data are global
                import mbpls_amd
                from mbpls_amd.comm import shard_columns
                comm, rank, world = shard
                offs = np.concatenate(([0], cuts, [p])).astype(np.int64)
                sizes = [int(offs[b + 1] - offs[b]) for b in range(nb)]
                ranges = shard_columns(sizes, rank, world)
                loc = lambda A: [np.ascontiguousarray(A[:, offs[b] + lo:offs[b] + hi]) for b, (lo, hi) in enumerate(ranges)]
                with mbpls_amd.column_sharded(comm):
                    m = est.fit(loc(X), Yin)
                    w = util.compare(util.model_to_flat(m), _local_view(ref, ranges, offs), tol_case, what=tag)
                    o = ors if sparse else orc
                    w = max(w, util.relerr(m.predict(loc(Xn)), o.predict(ref, sp(Xn))))
                    if nn > 1:
                        sgn = np.sign(np.sum(ref.Ts_ * m.Ts_, axis=0)); sgn[sgn == 0] = 1.0
                        w = max(w, util.relerr(m.transform(loc(Xn)) * sgn, o.transform(ref, sp(Xn))))
                    if not sparse and n >= 12 and do_cv:
                        # cross-validation under column shards: folds batched per pass (products and small sums
                        # all-reduced) against the fold-by-fold loop, every rank holding all predictions
                        from mbpls_amd.model_selection import cross_val_predict
                        folds = [3, 5, min(n, 40)][cv_pick]
                        est = MBPLS(n_components=min(K, 3), method=method, standardize=std, full_svd=True)
                        a_ = cross_val_predict(est, loc(X), Y, cv=folds)
                        b_ = cross_val_predict(est, loc(X), Y, cv=folds, batch_folds=False)
                        wcv = util.relerr(a_, b_)
                        if wcv > 1e-7:
                            print_(tag, f"cross_val_predict batched vs loop {wcv:.2e} (cv={folds}, rank {rank} of {world})")
                        w = max(w, wcv)
                if bound is not None:
                    assert w <= bound, f"{tag}: {w:.2e} beyond 50 x the oracle's own defect / one-ulp sensitivity {bound:.2e}"
                    print_(tag, f"relerr {w:.2e} within 50 x the oracle's own defect / one-ulp sensitivity {bound:.2e} (rank {rank} of {world})")
                    continue
                worst = max(worst, w)
                if w > 1e-9:
                    print_(tag, f"relerr {w:.2e} (rank {rank} of {world})")
                continue
            if form == 0 and not sparse:
                # a matrix already resident in HBM: fitted with copy=True, it must come back bit for bit
                import mbpls_amd
                from mbpls_amd import DeviceMatrix
                offs = np.concatenate(([0], cuts, [p])).astype(np.int64)
                dX = DeviceMatrix(mbpls_amd.get_context(), n, p, offs)
                dX.upload(X, 0)
                m = est.fit(dX, Yin)
                assert np.array_equal(dX.download(), X), "copy=True: the caller's resident matrix was modified"
                dX.close()
            elif form == 1:
                m = est.fit(sp(np.asfortranarray(X)) if nb == 1 else [np.asfortranarray(b) for b in sp(X)], Yin)
            elif form == 2 and not sparse:
                Xw = np.empty((n, 2 * p)); Xw[:, ::2] = X                    # non-contiguous views of a wider array
                m = est.fit(sp(Xw[:, ::2]) if nb == 1 else [Xw[:, ::2][:, a:b] for a, b in
                                                             zip(np.concatenate(([0], cuts)), np.concatenate((cuts, [p])))], Yin)
            else:
                m = est.fit(sp(X), Yin)
            w = util.compare(util.model_to_flat(m), util.model_to_flat(ref), tol_case, what=tag)
            o = ors if sparse else orc
            w = max(w, util.relerr(m.predict(sp(X)), o.predict(ref, sp(X))))
            # new data: predict, and transform with block scores (the reference has none for SIMPLS, nor -- a bug of its
            # own, SURVEY 8c' G5 -- without standardisation and 2-D Y)
            w = max(w, util.relerr(m.predict(sp(Xn)), o.predict(ref, sp(Xn))))
            blocks_ok = method != "SIMPLS"
            got = m.transform(sp(Xn), Yn, return_block_scores=blocks_ok)
            want = o.transform(ref, sp(Xn), Yn, return_block_scores=blocks_ok)
            sgn = np.sign(np.sum(ref.Ts_ * m.Ts_, axis=0)); sgn[sgn == 0] = 1.0
            if nn > 1:                                     # (one new sample: U of the transform is +-1 per column, sign only)
                w = max(w, util.relerr(got[0] * sgn, want[0]), util.relerr(got[-1] * sgn, want[-1]))
                if blocks_ok:
                    for a, b in zip(got[1], want[1]):
                        assert a.shape == b.shape, (a.shape, b.shape)      # (KERNEL, calc_all=False: no columns, both)
                        if a.shape[1]:
                            w = max(w, util.relerr(a * sgn, b))
            # now and then: cross-validation, folds batched per pass over X against the fold-by-fold loop (small shapes)
            if not (big or long or wide) and not sparse and n >= 12 and do_cv:
                from mbpls_amd.model_selection import cross_val_predict
                folds = [3, 5, min(n, 40)][cv_pick]
                est = MBPLS(n_components=min(K, 3), method=method, standardize=std, full_svd=True)
                a_ = cross_val_predict(est, sp(X), Y, cv=folds)
                b_ = cross_val_predict(est, sp(X), Y, cv=folds, batch_folds=False)
                wcv = util.relerr(a_, b_)
                if wcv > 1e-7:
                    print_(tag, f"cross_val_predict batched vs loop {wcv:.2e} (cv={folds})")
                w = max(w, wcv)
            if bound is not None:
                assert w <= bound, f"{tag}: {w:.2e} beyond 50 x the oracle's own defect / one-ulp sensitivity {bound:.2e}"
                print_(tag, f"relerr {w:.2e} within 50 x the oracle's own defect / one-ulp sensitivity {bound:.2e}")
            else:
                worst = max(worst, w)
                if w > 1e-9:
                    print_(tag, f"relerr {w:.2e}")
        except Exception as e:
            bad += 1
            print_(tag, "FAILED:", type(e).__name__, str(e)[:200])
            if raise_errors:
                raise
        finally:
            for key, (_, default) in tun.items():
                mbpls_amd.get_context().set_tunable(key, default)
            os.environ.pop("MBPLS_NIPALS_Q1_FIXED_POINT", None)
            for key in route:
                os.environ.pop(key, None)
            FitOptions.forced.clear()
    return bad, worst
