"""GPU parity for FEW samples (n <= 1024): csrc/colpass_sub.hip -- four columns per wave on 16-lane DPP rows (n <= 256),
two on half waves (n <= 512), one per wave without any barrier (n <= 1024).  Every flavour of the column pass has a counterpart there; each is checked against the
oracle and against the one-column-per-workgroup kernels (tunable sub_columns = 0) on the same data.
Tolerance 1e-6 relative, sign-aligned."""
import numpy as np
import pytest

from oracle import mbpls_oracle as orc
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-6


def _both(ctx, make):
    """fit with the sub-wave kernels and with the generic ones"""
    out = []
    for flag in (1, 0):
        ctx.set_tunable("sub_columns", flag)
        try:
            out.append(make())
        finally:
            ctx.set_tunable("sub_columns", 1)
    return out


@pytest.mark.parametrize("n", [17, 40, 64, 65, 90, 128, 129, 150, 192, 200, 256, 257, 300, 384, 511, 512, 513, 700, 768, 1024])
@pytest.mark.parametrize("method", ["NIPALS", "UNIPALS"])
def test_one_response_never_writes_x(gpu_required, method, n):
    """the device-resident queue: statistics-only first pass, lazy sweep, lazy loadings + sweep, loadings only"""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    p, K = 1500, 6
    X, Y = orc.make_latent(n, p, 1, seed=n)
    blocks = [X[:, :400], X[:, 400:]]
    ref = orc.fit(blocks, Y, n_components=min(K, n - 1), method=method)
    a, b = _both(ctx, lambda: MBPLS(n_components=min(K, n - 1), method=method).fit(blocks, Y))
    for mdl in (a, b):
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} n={n}")
        assert w < 1e-8, w
    assert util.compare(util.model_to_flat(a), util.model_to_flat(b), 1e-9, what="sub vs generic") < 1e-10


@pytest.mark.parametrize("n,K", [(100, 20), (300, 38), (600, 38)])
def test_many_lazy_corrections(gpu_required, n, K):
    """kprev beyond the lanes of a column group: a lane carries components s, s + 16, s + 32 (s, s + 32 on half waves)"""
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, 700))
    Y = X[:, :5].sum(axis=1, keepdims=True) + 0.1 * rng.standard_normal((n, 1))
    ref = orc.fit(X, Y, n_components=K, method="UNIPALS")
    a, b = _both(ctx, lambda: MBPLS(n_components=K, method="UNIPALS").fit(X, Y))
    for mdl in (a, b):
        assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"K={K}") < 1e-6
    # deep components of noise: R_ = W (P^T W)^-1 amplifies the last-bit differences of the two reduction orders
    assert util.compare(util.model_to_flat(a), util.model_to_flat(b), TOL, what="sub vs generic") < TOL


@pytest.mark.parametrize("n", [40, 90, 150, 400, 700, 900])
@pytest.mark.parametrize("env", [{"MBPLS_DEFLATION": "explicit", "MBPLS_IMPLICIT_STD": "0"},
                                 {"MBPLS_IMPLICIT_STD": "0"}, {"MBPLS_RESIDENT": "0"}])
def test_storing_and_host_driven_flavours(gpu_required, monkeypatch, n, env):
    """storing standardise (+ first sweep), explicit deflation (+ sweep), lazy passes without implicit standardisation,
    the host-driven loop; several responses (explicit passes, products) and no standardisation as well"""
    import mbpls_amd
    from mbpls_amd import MBPLS
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    ctx = mbpls_amd.get_context()
    X, Y = orc.make_latent(n, 900, 2, seed=3 * n)
    blocks = [X[:, :300], X[:, 300:]]
    for method, q, std in (("NIPALS", 1, True), ("NIPALS", 1, False), ("NIPALS", 2, True), ("UNIPALS", 2, True),
                           ("KERNEL", 1, True), ("SIMPLS", 2, True)):
        ref = orc.fit(blocks, Y[:, :q], n_components=4, method=method, standardize=std, full_svd=True)
        a, b = _both(ctx, lambda: MBPLS(n_components=4, method=method, standardize=std, full_svd=True).fit(blocks, Y[:, :q]))
        for mdl in (a, b):
            w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} q={q} std={std} n={n}")
            assert w < 1e-8, w


@pytest.mark.parametrize("n", [90, 350])
def test_offset_data_and_row_subsets(gpu_required, n):
    """centring flavour (x - mean in registers) and the row mask of cross-validation folds"""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    ctx = mbpls_amd.get_context()
    X, Y = orc.make_latent(n, 800, 1, seed=n)
    Xo = X + 500.0 * (1.0 + np.arange(800) % 3)
    ref = orc.fit(Xo, Y, n_components=3)
    a, b = _both(ctx, lambda: MBPLS(n_components=3).fit(Xo, Y))
    for mdl in (a, b):
        util.check_nipals_iterations(mdl, ref)
        assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="offset") < 1e-6
    mask = np.ones(n, dtype=bool)
    mask[::5] = False
    dX = DeviceMatrix(ctx, n, 800, [0, 300, 800])
    dX.upload(X, 0)
    refm = orc.fit([X[mask][:, :300], X[mask][:, 300:]], Y[mask], n_components=3)
    for mdl in _both(ctx, lambda: MBPLS(n_components=3)._fit(dX, Y, mask)):
        assert util.compare(util.model_to_flat(mdl), util.model_to_flat(refm), TOL, what="fold") < 1e-9
    dX.close()


@pytest.mark.parametrize("n,q", [(30, 1), (45, 1), (64, 2), (90, 1), (100, 1), (150, 1), (250, 2), (350, 1), (400, 1), (512, 1),
                                 (700, 1), (800, 1)])
def test_missing_values(gpu_required, n, q):
    """sparse_data=True: the masked passes (standardise + sweep, sweep, loadings + deflation + sweep) in their
    column-group form, against the NaN-aware oracle and the one-column-per-workgroup kernels; a fold of such data too"""
    import warnings

    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from oracle import mbpls_oracle_sparse as ors
    ctx = mbpls_amd.get_context()
    p, K = 1100, 3
    X, Y = orc.make_latent(n, p, q, seed=7 * n)
    rng = np.random.RandomState(n)
    X = X.copy()
    X[rng.rand(n, p) < 0.04] = np.nan
    blocks = [X[:, :500], X[:, 500:]]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = ors.fit(blocks, Y, n_components=K)
        a, b = _both(ctx, lambda: MBPLS(n_components=K, sparse_data=True).fit(blocks, Y))
    for mdl in (a, b):
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"sparse n={n}")
        assert w < 1e-8, w
        util.check_nipals_iterations(mdl, ref, two=False)
    assert util.compare(util.model_to_flat(a), util.model_to_flat(b), 1e-9, what="sub vs generic") < 1e-9
    if q == 1:
        mask = np.ones(n, dtype=bool)
        mask[1::4] = False
        dX = DeviceMatrix(ctx, n, p, [0, 500, p])
        dX.upload(X, 0)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            refm = ors.fit([X[mask][:, :500], X[mask][:, 500:]], Y[mask], n_components=2)
            folds = _both(ctx, lambda: MBPLS(n_components=2, sparse_data=True)._fit(dX, Y, mask))
        for mdl in folds:
            assert util.compare(util.model_to_flat(mdl), util.model_to_flat(refm), TOL, what="sparse fold") < 1e-8
        dX.close()


@pytest.mark.parametrize("n", [20, 64, 100, 256])
def test_predict_and_transform_products(gpu_required, n):
    """X W products (predict, transform, block scores) with the row pairs x column lanes mapping of k_xw_small"""
    import mbpls_amd
    from mbpls_amd import MBPLS
    from tests.test_gpu_parity import _fit_flat
    ctx = mbpls_amd.get_context()
    X, Y = orc.make_latent(n, 1300, 3, seed=n + 1)
    blocks = [X[:, :500], X[:, 500:]]
    Xn, Yn = orc.make_latent(n, 1300, 3, seed=n + 2)        # new data with as few samples
    bn = [Xn[:, :500], Xn[:, 500:]]
    for method in ("NIPALS", "KERNEL"):
        ref = orc.fit(blocks, Y, n_components=3, method=method)
        flat = util.model_to_flat(ref)
        flat["predict"] = orc.predict(ref, bn)
        Ts, T, U = orc.transform(ref, bn, Yn, return_block_scores=True)
        flat["Ts_new"], flat["U_new"] = Ts, U
        for b, blk in enumerate(T):
            flat[f"T_new/{b}"] = blk
        for mdl_flat in _both(ctx, lambda: _fit_flat(MBPLS(n_components=3, method=method).fit(blocks, Y), bn, Yn)):
            assert util.compare(mdl_flat, flat, TOL, what=f"{method} n={n}") < 1e-8
