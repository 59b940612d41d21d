"""GPU parity for sparse_data=True (missing values in X, NaN-aware NIPALS).

Reference behaviour: mbpls/mbpls.py:846-854 (weights), :865-873 (block scores), :919-926 (loadings),
:1376-1384 (predict), :1106-1153 (transform).  Checked against fixtures captured from the imported
reference (tests/golden/sparse.npz) and against the NaN-aware oracle on larger seeded problems.
Tolerance: 1e-6 relative, sign-aligned (north_star)."""
import warnings

import numpy as np
import pytest

from oracle import mbpls_oracle as orc
from oracle import mbpls_oracle_sparse as ors
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _flat(model, Xn, Yn):
    flat = util.model_to_flat(model)
    flat["predict"] = model.predict(Xn)
    Ts, T, U = model.transform(Xn, Yn, return_block_scores=True)
    flat["Ts_new"], flat["U_new"] = Ts, U
    for b, blk in enumerate(T):
        flat[f"T_new/{b}"] = blk
    return flat


def _oracle_flat(ref, Xn, Yn):
    flat = util.model_to_flat(ref)
    flat["predict"] = ors.predict(ref, Xn)
    Ts, T, U = ors.transform(ref, Xn, Yn, return_block_scores=True)
    flat["Ts_new"], flat["U_new"] = Ts, U
    for b, blk in enumerate(T):
        flat[f"T_new/{b}"] = blk
    return flat


def test_reference_fixture(gpu_required):
    from mbpls_amd import MBPLS
    g = util.load("sparse")
    worst = 0.0
    for q in (1, 2):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        Xn, Yn = g[f"in/q{q}/Xnew"], g[f"in/q{q}/Ynew"]
        for blocks in ("single", "multi"):
            sp = (lambda A: A) if blocks == "single" else (lambda A: [np.ascontiguousarray(A[:, :40]),
                                                                       np.ascontiguousarray(A[:, 40:])])
            for std in (True, False):
                ref = util.sub(g, f"ref/q{q}/{blocks}/std{int(std)}")
                mdl = MBPLS(n_components=3, sparse_data=True, standardize=std).fit(sp(X), Y)
                worst = max(worst, util.compare(_flat(mdl, sp(Xn), Yn), ref, TOL, what=f"q{q} {blocks} std{std}"))
                ok = ~np.isnan(X)
                if std:
                    seen = np.concatenate([s.n_samples_seen_ for s in mdl.x_scalers_])
                    assert np.array_equal(seen, ok.sum(axis=0))
                # check_sparsity_level bookkeeping (mbpls.py:255-271)
                for b, blk in enumerate(sp(ok) if blocks == "multi" else [ok]):
                    sr, sc, dr, dc = mdl.sparse_X_info_[b]
                    assert np.array_equal(sr, np.where(~blk.all(axis=1))[0])
                    assert np.array_equal(sc, np.where(~blk.all(axis=0))[0])
                    assert np.array_equal(dr, np.where(blk.all(axis=1))[0])
                    assert np.array_equal(dc, np.where(blk.all(axis=0))[0])
                assert all(len(v) == 0 for v in mdl.sparse_Y_info_['Y'][:2])
    assert worst < 1e-8, worst


@pytest.mark.parametrize("n,p,q,K,blocks,frac", [
    (501, 1500, 2, 3, [500, 400, 600], 0.03),     # odd n: padded columns, several blocks
    (4096, 6000, 1, 3, [6000], 0.01),               # the n = 4096 configuration of the dense hot path
    (1000, 2000, 3, 3, [2000], 0.10),
    (9000, 700, 1, 2, [300, 400], 0.02),            # n > 8192: long-column kernels
])
def test_medium_vs_oracle(gpu_required, n, p, q, K, blocks, frac):
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(n, p, q, seed=n + p)
    Xn, Yn = orc.make_latent(65, p, q, seed=2)
    rng = np.random.RandomState(n)
    X = X.copy(); X[rng.rand(n, p) < frac] = np.nan
    Xn = Xn.copy(); Xn[rng.rand(65, p) < frac] = np.nan
    Xn[3] = orc.make_latent(65, p, q, seed=2)[0][3]      # a complete row among the sparse ones
    cuts = np.cumsum(blocks)[:-1]
    sp = (lambda A: A) if len(blocks) == 1 else (lambda A: [np.ascontiguousarray(b) for b in np.split(A, cuts, axis=1)])
    ref = ors.fit(sp(X), Y, n_components=K)
    mdl = MBPLS(n_components=K, sparse_data=True).fit(sp(X), Y)
    w = util.compare(_flat(mdl, sp(Xn), Yn), _oracle_flat(ref, sp(Xn), Yn), TOL, what=f"sparse n={n} p={p}")
    assert w < 1e-8, w
    util.check_nipals_iterations(mdl, ref, two=False)


def test_missing_responses_reference_fixture(gpu_required):
    """sparse_data=True with NaN in Y as well (mbpls.py:832-836 start vector = first complete response, :891-897
    masked Y loadings, :900-906 masked Y scores; transform :1156-1166), against attributes captured from the imported
    reference (tests/golden/sparse_y.npz; first response column sparse, q = 2 and 3, single / two blocks, +-std)."""
    from mbpls_amd import MBPLS
    g = util.load("sparse_y")
    worst = 0.0
    for q in (2, 3):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        Xn, Yn = g[f"in/q{q}/Xnew"], g[f"in/q{q}/Ynew"]
        for blocks in ("single", "multi"):
            sp = (lambda A: A) if blocks == "single" else (lambda A: [np.ascontiguousarray(A[:, :40]),
                                                                       np.ascontiguousarray(A[:, 40:])])
            for std in (True, False):
                ref = util.sub(g, f"ref/q{q}/{blocks}/std{int(std)}")
                Yc = Y.copy()
                mdl = MBPLS(n_components=3, sparse_data=True, standardize=std).fit(sp(X), Yc)
                assert np.array_equal(np.isnan(Yc), np.isnan(Y))                 # copy=True: the caller's Y is intact
                worst = max(worst, util.compare(_flat(mdl, sp(Xn), Yn), ref, TOL, what=f"q{q} {blocks} std{std}"))
                if std:
                    assert util.relerr(mdl.y_scaler_.mean_, ref["y_mean"]) < 1e-13
                    assert util.relerr(mdl.y_scaler_.scale_, ref["y_scale"]) < 1e-13
                    assert np.array_equal(mdl.y_scaler_.n_samples_seen_, (~np.isnan(Y)).sum(axis=0))
                sr, sc, dr, dc = mdl.sparse_Y_info_['Y']                          # check_sparsity_level (:295-296)
                assert np.array_equal(sr, np.where(np.isnan(Y).any(axis=1))[0])
                assert np.array_equal(sc, np.where(np.isnan(Y).any(axis=0))[0])
    assert worst < 1e-8, worst


@pytest.mark.parametrize("n,p,q,K,blocks", [(500, 1500, 3, 3, [700, 800]), (2048, 3000, 2, 2, [3000])])
def test_missing_responses_medium_vs_oracle(gpu_required, n, p, q, K, blocks):
    """missing responses at medium size: missing values in X (masked column passes on the GPU) and in Y (host, n x q)
    -- in rows the reference patches, so the oracle can follow.  One more case puts a missing response into a row
    WITHOUT missing X values, where the reference fails and this implementation regresses the row over the present
    responses like every other: the fit stays finite and close to the fit on the completed data."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(n, p, q, seed=n + q)
    rng = np.random.RandomState(n)
    X = X.copy(); X[rng.rand(n, p) < 0.0008] = np.nan
    cuts = np.cumsum(blocks)[:-1]
    sp = (lambda A: A) if len(blocks) == 1 else (lambda A: [np.ascontiguousarray(b) for b in np.split(A, cuts, axis=1)])
    last = sp(X)[-1] if len(blocks) > 1 else X
    rows = np.where(np.isnan(last).any(axis=1))[0]
    dense_rows = np.where(~np.isnan(X).any(axis=1))[0]
    assert rows.size > 40 and dense_rows.size > 5
    Y = Y.copy()
    Y[rows[:25], 0] = np.nan
    Y[rows[15:40], q - 1] = np.nan
    ref = ors.fit(sp(X), Y, n_components=K)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mdl = MBPLS(n_components=K, sparse_data=True).fit(sp(X), Y)
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"missing responses n={n}")
    assert w < 1e-7, w
    # dozens of iterations, the last ones within rounding of the 1e-14 stop rule (and the oracle's threaded BLAS does
    # not sum in a fixed order): the counts may differ by a few from run to run
    assert all(abs(a - b) <= max(5, 0.35 * b) for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_)), \
        (mdl.nipals_iterations_, ref.nipals_iterations_)
    # (in a response column that is sparse already: with no complete column left the start vector of every component
    # would be np.random.rand, mbpls.py:833-834, and the weakest component's fixed point would depend on the draw)
    Y2 = Y.copy(); Y2[dense_rows[:3], q - 1] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m2 = MBPLS(n_components=K, sparse_data=True).fit(sp(X), Y2)
    assert np.isfinite(m2.beta_).all() and np.isfinite(m2.Ts_).all()
    assert util.relerr(m2.beta_, mdl.beta_) < 0.05


def test_complete_data_equals_dense_fit(gpu_required):
    """Without a single NaN the masked passes must give the dense model."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(300, 2000, 2, seed=5)
    Xs = [np.ascontiguousarray(X[:, :900]), np.ascontiguousarray(X[:, 900:])]
    a = MBPLS(n_components=3).fit(Xs, Y)
    b = MBPLS(n_components=3, sparse_data=True).fit(Xs, Y)
    w = util.compare(util.model_to_flat(b), util.model_to_flat(a), 1e-10, what="complete")
    assert w < 1e-11, w
    assert np.allclose(a.predict(Xs), b.predict(Xs), rtol=0, atol=1e-11)


def test_method_is_switched_to_nipals(gpu_required):
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(60, 120, 1, seed=9)
    X = X.copy(); X[4, 7] = np.nan
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        mdl = MBPLS(n_components=2, method="KERNEL", sparse_data=True).fit(X, Y)      # mbpls.py:285-289
    assert mdl.method == "NIPALS" and any("NIPALS" in str(r.message) for r in rec)
    ref = ors.fit(X, Y, n_components=2)
    assert util.relerr(mdl.beta_, ref.beta_) < 1e-9
    Yb = Y.copy(); Yb[0, 0] = np.inf
    with pytest.raises(ValueError):
        MBPLS(n_components=2, sparse_data=True).fit(X, Yb)
    # and without sparse_data a NaN is still an input error
    with pytest.raises(ValueError):
        MBPLS(n_components=2).fit(X, Y)


def test_high_sparsity_warning(gpu_required):
    """mbpls.py:265-270: more than half of the rows (or columns) with missing entries -> UserWarning."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(50, 80, 1, seed=3)
    X = X.copy()
    X[:30, 0] = np.nan
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        MBPLS(n_components=2, sparse_data=True, standardize=False).fit(X, Y)
    assert any("sparsity" in str(r.message) for r in rec)


def test_row_subset_fit_and_cross_validation_with_missing_values(gpu_required):
    """A fold of data with missing values: the matrix stays resident, the held-out samples become zero rows of the
    working copy and count in no statistic (mbpls_hip_nan_standardize_sweep_masked).  Equal to the oracle fitted on
    the sliced arrays; cross_val_predict equals a loop of oracle fits."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix, cross_val_predict
    n, p, q, K = 420, 1500, 2, 3
    X, Y = orc.make_latent(n, p, q, seed=12)
    rng = np.random.RandomState(3)
    X = X.copy(); X[rng.rand(n, p) < 0.02] = np.nan
    Xs = [np.ascontiguousarray(X[:, :600]), np.ascontiguousarray(X[:, 600:])]
    ctx = mbpls_amd.get_context()
    Xd = DeviceMatrix(ctx, n, p, [0, 600, p])
    Xd.upload(Xs[0], 0); Xd.upload(Xs[1], 600)
    mask = np.ones(n, dtype=bool); mask[rng.choice(n, 90, replace=False)] = False
    ref = ors.fit([b[mask] for b in Xs], Y[mask], n_components=K)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mdl = MBPLS(n_components=K, sparse_data=True)._fit(Xd, Y, mask)
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="sparse row subset")
    assert w < 1e-8, w
    util.check_nipals_iterations(mdl, ref, two=False)
    assert np.array_equal(np.concatenate([s.n_samples_seen_ for s in mdl.x_scalers_]), (~np.isnan(X[mask])).sum(axis=0))
    sr, sc, dr, dc = mdl.sparse_X_info_[0]
    ok0 = ~np.isnan(Xs[0][mask])
    assert np.array_equal(sr, np.where(~ok0.all(axis=1))[0]) and np.array_equal(sc, np.where(~ok0.all(axis=0))[0])
    assert util.relerr(mdl.predict(Xd)[~mask], ors.predict(ref, [b[~mask] for b in Xs])) < 1e-8
    # cross-validation of the sparse estimator on the resident matrix
    folds = [(np.where(np.arange(n) % 3 != f)[0], np.where(np.arange(n) % 3 == f)[0]) for f in range(3)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = cross_val_predict(MBPLS(n_components=K, sparse_data=True), Xd, Y, cv=folds)
    want = np.empty_like(Y)
    for tr, te in folds:
        m = ors.fit([b[tr] for b in Xs], Y[tr], n_components=K)
        want[te] = ors.predict(m, [b[te] for b in Xs])
    assert util.relerr(got, want) < 1e-8
    assert np.isnan(Xd.download(0, 600)).sum() == np.isnan(Xs[0]).sum()      # the resident matrix is untouched
    Xd.close()


@pytest.mark.parametrize("n,q", [(32768, 1), (32770, 1), (40000, 3), (65536, 1)])
def test_missing_values_on_very_long_columns(gpu_required, n, q):
    """32768 < n <= 65536: the missing-value passes hold 64 rows per thread; the per-row presence flags need a 64-bit
    mask there (a 32-bit one gave rows 32..63 the flags of rows 0..31 -- NaN sums; found by the long-column fuzz sweep of
    round 5).  n = 32768 is the last shape of the 32-rows-per-thread configuration."""
    from mbpls_amd import MBPLS
    p = 22
    X, Y = orc.make_latent(n, p, q, r=4, seed=n)
    rng = np.random.RandomState(1)
    X[rng.rand(n, p) < 0.05] = np.nan
    blocks = [X[:, :10], X[:, 10:]]
    from mbpls_amd import get_context
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = ors.fit(blocks, Y, n_components=3)
    # the register shapes ("nan_long_from" = 65536; no longer the default above 8192 samples: they spill), the default
    # (two streaming phases) and the one-workgroup-per-chunk streamed kernel
    for frm, two_phase in ((65536, 1), (8192, 1), (8192, 0)):
        get_context().set_tunable("nan_long_from", frm)
        get_context().set_tunable("nan_long_two_phase", two_phase)
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                mdl = MBPLS(n_components=3, sparse_data=True).fit(blocks, Y)
        finally:
            get_context().set_tunable("nan_long_from", 8192)
            get_context().set_tunable("nan_long_two_phase", 1)
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"sparse n={n} q={q}")
        assert w < 1e-8, w
        if q == 1:
            util.check_nipals_iterations(mdl, ref, two=False)
        assert util.relerr(mdl.predict(blocks), ors.predict(ref, blocks)) < 1e-8


@pytest.mark.parametrize("n,q,std", [(65538, 1, True), (70001, 2, True), (131072, 1, False)])
def test_missing_values_beyond_the_register_shapes(gpu_required, n, q, std):
    """n > 65536: no workgroup holds a column any more -- the missing-value passes stream it (k_colpass_nan_long), any n;
    same sums in the same places (:846-873, :919-926, :969)."""
    from mbpls_amd import MBPLS
    p = 22
    X, Y = orc.make_latent(n, p, q, r=4, seed=n)
    rng = np.random.RandomState(2)
    X[rng.rand(n, p) < 0.04] = np.nan
    blocks = [X[:, :9], X[:, 9:]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = ors.fit(blocks, Y, n_components=3, standardize=std)
        mdl = MBPLS(n_components=3, sparse_data=True, standardize=std).fit(blocks, Y)
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"sparse n={n} q={q}")
    assert w < 1e-8, w
    if q == 1:
        util.check_nipals_iterations(mdl, ref, two=False)
    else:     # (dozens of iterations towards 1e-14 on 70 001-term sums: the last one is a matter of rounding)
        assert all(abs(a - b) <= 1 for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_))
    Xn = [b[:50] for b in blocks]
    assert util.relerr(mdl.predict(Xn), ors.predict(ref, Xn)) < 1e-8
    assert util.relerr(mdl.transform(Xn), ors.transform(ref, Xn)) < 1e-8


@pytest.mark.parametrize("two_phase", [1, 0])
@pytest.mark.parametrize("n,p,q,std", [(300, 40, 1, True), (1031, 64, 3, True), (4096, 30, 1, False), (5000, 200, 2, True),
                                       (2049, 3000, 1, True), (1500, 77, 2, False)])
def test_streamed_missing_value_passes_at_every_n(gpu_required, n, p, q, std, two_phase):
    """the long-column missing-value passes at sizes the register shapes cover ("nan_long_from" = 0) -- the two streaming
    phases of colpass_nan_long.hip (default) and the one-workgroup-per-chunk kernel they replaced ("nan_long_two_phase"
    = 0): equal to the oracle, and to the register-resident passes to rounding; cross-validation folds (row masks)
    included"""
    from mbpls_amd import MBPLS, get_context
    from mbpls_amd.model_selection import cross_val_predict
    X, Y = orc.make_latent(n, p, q, r=4, seed=n + p)
    rng = np.random.RandomState(3)
    X[rng.rand(n, p) < 0.05] = np.nan
    X[:, 3] += 7.0
    blocks = [X[:, :p // 3], X[:, p // 3:]]
    ctx = get_context()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = ors.fit(blocks, Y, n_components=3, standardize=std)
        est = MBPLS(n_components=3, sparse_data=True, standardize=std)
        reg = MBPLS(n_components=3, sparse_data=True, standardize=std).fit(blocks, Y)
        cv_reg = cross_val_predict(est, blocks, Y, cv=3) if n <= 1100 else None
        ctx.set_tunable("nan_long_from", 0)
        ctx.set_tunable("nan_long_two_phase", two_phase)
        try:
            mdl = MBPLS(n_components=3, sparse_data=True, standardize=std).fit(blocks, Y)
            cv_str = cross_val_predict(est, blocks, Y, cv=3) if n <= 1100 else None
        finally:
            ctx.set_tunable("nan_long_from", 8192)
            ctx.set_tunable("nan_long_two_phase", 1)
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"streamed sparse n={n}")
    assert w < 1e-8, w
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(reg), TOL, what="streamed vs register") < 1e-9
    if q == 1:
        util.check_nipals_iterations(mdl, ref, two=False)
    else:     # (another summation order: the last of dozens of iterations towards 1e-14 is a matter of rounding)
        assert all(abs(a - b) <= 1 for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_))
    if cv_reg is not None:
        assert util.relerr(cv_str, cv_reg) < 1e-9
