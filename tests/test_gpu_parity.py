"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the golden
fixtures captured from the reference.  Tolerance: north_star's 1e-6 relative (float64,
sign-aligned); the kernel-level checks use tighter bounds to catch indexing errors early."""
import warnings

import numpy as np
import pytest

from oracle import mbpls_oracle as orc
from tests import util

pytestmark = pytest.mark.gpu

TOL = 1e-6          # BASELINE.json north_star: 1e-6 relative float64, sign-aligned

_ORACLE_CACHE = {}


def _cached_oracle(key, make):
    """several-response oracle fits take seconds of CPU: parametrisations that differ only in the GPU route share one"""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = make()
    return _ORACLE_CACHE[key]
TIGHT = 1e-11       # what we actually expect (summation-order differences only)


@pytest.fixture(scope="module")
def ctx(gpu_required):
    import mbpls_amd
    return mbpls_amd.get_context()


# ------------------------------------------------------------------ kernel level
@pytest.mark.parametrize("n,p,order", [(40, 70, "C"), (41, 33, "C"), (129, 257, "F"), (512, 300, "C"),
                                       (1000, 64, "C"), (2048, 96, "F"), (4096, 40, "C"), (8192, 24, "C")])
def test_upload_download_roundtrip(ctx, n, p, order):
    from mbpls_amd import DeviceMatrix
    rng = np.random.RandomState(n + p)
    X = np.asarray(rng.standard_normal((n, p)), order=order)
    m = DeviceMatrix(ctx, n, p)
    m.upload(X)
    assert np.array_equal(m.download(order="C"), X)
    assert np.array_equal(m.download(order="F"), X)
    # column sub-range and a strided view (column slice of a wider C array)
    wide = rng.standard_normal((n, p + 7))
    m.upload(wide[:, 3:3 + p])
    assert np.array_equal(m.download(2, p - 2), wide[:, 5:3 + p])
    m.close()


@pytest.mark.parametrize("n,p,blocks", [(40, 70, [25, 45]), (41, 33, [33]), (300, 500, [100, 150, 250]),
                                        (1000, 257, [257]), (2048, 400, [130, 270]), (4096, 300, [300]),
                                        (8192, 100, [40, 60]), (9001, 48, [20, 28]), (33000, 24, [24])])
def test_column_pass_kernels(ctx, n, p, blocks):
    """standardize_sweep / sweep / deflate_sweep against numpy on the same data."""
    from mbpls_amd import DeviceMatrix, _capi
    rng = np.random.RandomState(7 * n + p)
    X = rng.standard_normal((n, p)) * rng.uniform(0.5, 3.0, p) + rng.uniform(-2, 2, p)
    X[:, 3] = 1.75                                   # constant column -> scale 1
    offs = np.concatenate(([0], np.cumsum(blocks)))
    B = len(blocks)
    v = rng.standard_normal(n)
    m = DeviceMatrix(ctx, n, p, offs)
    m.upload(X)
    d_mean, d_scale = ctx.empty(p), ctx.empty(p)
    d_ssqx, d_ssqg, d_ssqp = ctx.empty(B), ctx.empty(B), ctx.empty(B)
    d_g, d_t, d_p = ctx.empty(p), ctx.empty(n * B), ctx.empty(p)
    d_v, d_ts = ctx.from_host(v), ctx.empty(n)
    alpha = 1.0 / v.dot(v)
    m.standardize_sweep(m, _capi.SCALE_FIT, d_mean, d_scale, d_ssqx, d_v, alpha, d_g, d_t, d_ssqg)
    sc = orc.ColumnScaler().fit(X)
    Z = sc.transform(X)
    assert util.relerr(d_mean.get(), sc.mean_) < 1e-14
    assert util.relerr(d_scale.get(), sc.scale_) < 1e-14
    assert d_scale.get()[3] == 1.0
    Zd = m.download()
    assert util.relerr(Zd, Z) < 1e-13
    g = alpha * Z.T.dot(v)
    rng_b = list(zip(offs[:-1], offs[1:]))
    assert util.relerr(d_g.get(), g) < TIGHT
    t_ref = np.concatenate([Z[:, s:e].dot(g[s:e]) for s, e in rng_b])
    assert util.relerr(d_t.get(), t_ref) < TIGHT
    assert util.relerr(d_ssqg.get(), [g[s:e].dot(g[s:e]) for s, e in rng_b]) < TIGHT
    assert util.relerr(d_ssqx.get(), [(Z[:, s:e] ** 2).sum() for s, e in rng_b]) < TIGHT
    # plain sweep reproduces the fused one bit for bit (same kernel math, same chunking)
    d_g2, d_t2, d_s2 = ctx.empty(p), ctx.empty(n * B), ctx.empty(B)
    m.sweep(d_v, alpha, d_g2, d_t2, d_s2)
    assert util.relerr(d_t2.get(), t_ref) < TIGHT
    m.sweep(d_v, alpha, d_g, d_t, d_ssqg)            # determinism: two identical launches
    assert np.array_equal(d_t.get(), d_t2.get()) and np.array_equal(d_g.get(), d_g2.get())
    # deflate + fused next sweep
    ts = rng.standard_normal(n)
    ts /= np.linalg.norm(ts)
    d_ts.set(ts)
    m.deflate_sweep(d_ts, 1.0, True, d_p, d_ssqp, d_v, alpha, d_g, d_t, d_ssqg)
    pl = Z.T.dot(ts)
    Zn = Z - np.outer(ts, pl)
    assert util.relerr(d_p.get(), pl) < TIGHT
    assert util.relerr(m.download(), Zn) < 1e-12
    g = alpha * Zn.T.dot(v)
    assert np.max(np.abs(d_g.get() - g)) < 1e-11 * max(1.0, np.max(np.abs(alpha * Z.T.dot(v))))
    t_ref = np.concatenate([Zn[:, s:e].dot(g[s:e]) for s, e in rng_b])
    assert np.max(np.abs(d_t.get() - t_ref)) < 1e-10 * np.max(np.abs(t_ref))
    assert util.relerr(d_ssqp.get(), [pl[s:e].dot(pl[s:e]) for s, e in rng_b]) < TIGHT
    # loadings only (last component): X untouched
    ts2 = rng.standard_normal(n)
    d_ts.set(ts2)
    m.deflate_sweep(d_ts, 2.0, False, d_p, d_ssqp)
    assert util.relerr(d_p.get(), 2.0 * Zn.T.dot(ts2)) < TIGHT
    assert util.relerr(m.download(), Zn) < 1e-12
    # standardise with given statistics (predict/transform path) and the no-scaling mode
    m2 = DeviceMatrix(ctx, n, p, offs)
    m2.upload(X)
    m2.standardize_sweep(m2, _capi.SCALE_APPLY, d_mean, d_scale, d_ssqx)
    assert util.relerr(m2.download(), Z) < 1e-13
    m2.upload(X)
    m2.standardize_sweep(m2, _capi.SCALE_NONE, None, None, d_ssqx)
    assert np.array_equal(m2.download(), X)
    assert util.relerr(d_ssqx.get(), [(X[:, s:e] ** 2).sum() for s, e in rng_b]) < TIGHT
    m.close(); m2.close()


@pytest.mark.parametrize("n,p,c,blocks", [(40, 70, 1, [25, 45]), (41, 33, 3, [33]), (300, 500, 5, [100, 400]),
                                          (1030, 257, 9, [257]), (4096, 300, 2, [100, 200]), (8192, 64, 4, [64]),
                                          # c > 8: x_w on the matrix cores (odd n, ragged blocks, > 64 rhs)
                                          (41, 33, 12, [33]), (517, 1001, 20, [3, 498, 500]), (4096, 700, 45, [700]),
                                          (300, 90, 70, [40, 50])])
def test_multi_rhs_products(ctx, n, p, c, blocks):
    from mbpls_amd import DeviceMatrix
    rng = np.random.RandomState(n + p + c)
    X = rng.standard_normal((n, p))
    offs = np.concatenate(([0], np.cumsum(blocks)))
    B = len(blocks)
    m = DeviceMatrix(ctx, n, p, offs)
    m.upload(X)
    V = rng.standard_normal((n, c))
    W = rng.standard_normal((p, c))
    dV, dW = ctx.from_host(V), ctx.from_host(W)
    dG, dssq = ctx.empty((p, c)), ctx.empty((B, c))
    m.xt_v(dV, c, 0.5, dG, dssq)
    G = 0.5 * X.T.dot(V)
    assert util.relerr(dG.get(), G) < TIGHT
    assert util.relerr(dssq.get(), [(G[s:e] ** 2).sum(axis=0) for s, e in zip(offs[:-1], offs[1:])]) < TIGHT
    out = ctx.empty((n, c))
    m.x_w(dW, c, out)
    assert util.relerr(out.get(), X.dot(W)) < TIGHT
    outb = ctx.empty(n * c * B)
    m.x_w(dW, c, outb, per_block=True)
    h = outb.get()
    for b, (s, e) in enumerate(zip(offs[:-1], offs[1:])):
        assert util.relerr(h[b * n * c:(b + 1) * n * c].reshape(c, n).T, X[:, s:e].dot(W[s:e])) < TIGHT
    mu, sc = rng.standard_normal(p), rng.uniform(0.5, 2.0, p)
    m.x_w(dW, c, out, mean=ctx.from_host(mu), scale=ctx.from_host(sc))
    assert util.relerr(out.get(), ((X - mu) / sc).dot(W)) < TIGHT
    m.close()
    # missing values: NaN contributes nothing / presence-indicator sums (mbpls_hip_x_w_nan), per block
    Xn = X.copy()
    Xn[rng.rand(n, p) < 0.1] = np.nan
    mn = DeviceMatrix(ctx, n, p, offs)
    mn.upload(Xn)
    ok = ~np.isnan(Xn)
    mn.x_w_nan(dW, c, out, mean=ctx.from_host(mu), scale=ctx.from_host(sc))
    assert util.relerr(out.get(), np.where(ok, (Xn - mu) / sc, 0.0).dot(W)) < TIGHT
    mn.x_w_nan(dW, c, outb, per_block=True, indicator=True)
    h = outb.get()
    for b, (s, e) in enumerate(zip(offs[:-1], offs[1:])):
        assert util.relerr(h[b * n * c:(b + 1) * n * c].reshape(c, n).T, ok[:, s:e].astype(float).dot(W[s:e])) < TIGHT
    mn.close()


@pytest.mark.parametrize("n,p", [(40, 200), (129, 1000), (256, 515), (1000, 3001), (2048, 4100)])
def test_gram(ctx, n, p):
    from mbpls_amd import DeviceMatrix
    rng = np.random.RandomState(n)
    X = rng.standard_normal((n, p))
    m = DeviceMatrix(ctx, n, p)
    m.upload(X)
    ldg = n + (n & 1)
    G = ctx.empty((ldg, n), zero=True)
    m.gram(G, ldg)
    Gh = G.get()[:n, :n]          # column-major (ldg x n): symmetric, so orientation is moot
    assert np.array_equal(Gh, Gh.T)
    assert util.relerr(Gh, X.dot(X.T)) < TIGHT
    m.close()


def test_fill_latent_statistics(ctx):
    from mbpls_amd import DeviceMatrix
    n, p, r = 512, 4096, 8
    rng = np.random.RandomState(0)
    Ts = rng.standard_normal((n, r)) * 0.85 ** np.arange(r)
    m = DeviceMatrix(ctx, n, p)
    m.fill_latent(ctx.from_host(Ts), r, 0.5, 1234, 0)
    X = m.download()
    m2 = DeviceMatrix(ctx, n, p // 2)
    m2.fill_latent(ctx.from_host(Ts), r, 0.5, 1234, p // 2)     # same global columns from another "rank"
    assert np.array_equal(m2.download(), X[:, p // 2:])
    # residual after projecting on the planted scores is the N(0, 0.25) noise
    Q, _ = np.linalg.qr(Ts)
    R = X - Q.dot(Q.T.dot(X))
    assert abs(R.var() / 0.25 * n / (n - r) - 1.0) < 0.02
    assert abs(np.mean(X)) < 0.02
    m.close(); m2.close()


# ------------------------------------------------------------------ estimator level
def _fit_flat(model, Xn=None, Yn=None, block_scores=True):
    flat = util.model_to_flat(model)
    if Xn is not None:
        flat["predict"] = model.predict(Xn)
        if model.method != "SIMPLS" and block_scores:
            Ts, T, U = model.transform(Xn, Yn, return_block_scores=True)
            for b, blk in enumerate(T):
                flat[f"T_new/{b}"] = blk
        else:
            Ts, U = model.transform(Xn, Yn)
        flat["Ts_new"], flat["U_new"] = Ts, U
    return flat


@pytest.mark.parametrize("fixture,methods", [("ref_golden_pgtn", ["UNIPALS", "NIPALS", "KERNEL", "SIMPLS"]),
                                             ("ref_golden_ngtp", ["UNIPALS", "KERNEL"])])
def test_reference_golden_csvs(gpu_required, kernel_route, fixture, methods):
    """The reference's own test (mbpls/tests/test_mbpls.py:34-421) on the HIP path (KERNEL p > n on both of its routes)."""
    from mbpls_amd import MBPLS
    g = util.load(fixture)
    X, Xt = [g["x1_train"], g["x2_train"]], [g["x1_test"], g["x2_test"]]
    preds = []
    for m in methods:
        mdl = MBPLS(n_components=2, method=m, standardize=True, full_svd=True).fit(X, g["y_train"])
        csv = util.sub(g, f"csv/{m}")
        got = {"Ts": mdl.Ts_, "U": mdl.U_, "V": mdl.V_, "beta": mdl.beta_, "P1": mdl.P_[0], "P2": mdl.P_[1],
               "Y_predict_test": mdl.predict(Xt)}
        if m != "SIMPLS":
            Ts_t, T_t, U_t = mdl.transform(Xt, g["y_test"], return_block_scores=True)
            got.update(T=np.hstack(mdl.T_), A=mdl.A_, T_test=np.hstack(T_t))
        else:
            Ts_t, U_t = mdl.transform(Xt, g["y_test"])
        got.update(Ts_test=Ts_t, U_test=U_t)
        assert set(got) == set(csv)
        for k, ref in csv.items():
            assert np.allclose(np.abs(got[k]), np.abs(ref)), (m, k)      # the reference's own criterion
            assert util.relerr(np.abs(got[k]), np.abs(ref)) < TOL, (m, k)
        util.compare(_fit_flat(mdl, Xt, g["y_test"]), util.sub(g, f"ref/{m}/full1"), TOL, what=f"{fixture}/{m}")
        preds.append(got["Y_predict_test"])
    for p in preds:
        assert np.allclose(preds[0], p, atol=1e-3)


def test_quickstart_config1(gpu_required, q1_route, kernel_route):
    """BASELINE.json configs[0]: README quick-start, n=40 p=200 K=2 (one-response NIPALS and KERNEL on both of their
    routes)."""
    from mbpls_amd import MBPLS
    g = util.load("quickstart_c1")
    for m in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
        mdl = MBPLS(n_components=2, method=m).fit(g["x"], g["y"])
        w = util.compare(_fit_flat(mdl, g["x"], g["y"]), util.sub(g, f"ref/{m}"), TOL, what=m)
        assert w < 1e-9
        if m == "NIPALS":
            assert mdl.nipals_iterations_ == ([1, 1] if q1_route == "one_pass" else [2, 2])


@pytest.mark.parametrize("fixture", ["matrix_pgtn", "matrix_ngtp"])
def test_matrix(gpu_required, monkeypatch, fixture):
    from mbpls_amd import MBPLS
    g = util.load(fixture)
    splits = np.cumsum(g["splits"])[:-1]
    worst = 0.0
    for q in (1, 3):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        Xn, Yn = g[f"in/q{q}/Xnew"], g[f"in/q{q}/Ynew"]
        for blocks in ("single", "multi"):
            sp = (lambda A: A) if blocks == "single" else (lambda A: list(np.split(A, splits, axis=1)))
            for m in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
                for std in (True, False):
                    for ca in (True, False):
                        pre = f"ref/q{q}/{blocks}/{m}/std{int(std)}/ca{int(ca)}"
                        ref = util.sub(g, pre)
                        # (KERNEL p > n: on the association matrix, then on the maintained thin products)
                        for kg in (("1", "0") if (m == "KERNEL" and fixture == "matrix_pgtn") else ("1",)):
                            monkeypatch.setenv("MBPLS_KERNEL_GRAM", kg)
                            mdl = MBPLS(n_components=3, method=m, standardize=std, calc_all=ca).fit(sp(X), Y)
                            if m == "KERNEL" and fixture == "matrix_pgtn":
                                assert ("matrix cores" in mdl.fit_plan_["solver"]) == (kg == "1"), mdl.fit_plan_
                            has_T_new = any(k.startswith("T_new") for k in ref) and (m != "KERNEL" or ca)
                            got = _fit_flat(mdl, sp(Xn), Yn, block_scores=has_T_new)
                            skip = () if has_T_new else ("T_new",)
                            worst = max(worst, util.compare(got, ref, TOL, skip=skip, what=pre + " gram=" + kg))
    assert worst < 1e-8


def test_edges(gpu_required):
    from mbpls_amd import MBPLS
    g = util.load("edges")
    mdl = MBPLS(n_components=3).fit(g["const/X"], g["const/Y"])
    assert mdl.x_scalers_[0].scale_[5] == 1.0 and mdl.beta_[5, 0] == 0.0
    util.compare(_fit_flat(mdl, g["const/X"], g["const/Y"]), util.sub(g, "ref/const"), TOL, what="const")
    X, Y = g["plain/X"], g["plain/Y"]
    m0 = MBPLS(n_components=3).fit(X, Y)
    m1 = MBPLS(n_components=3).fit(X, Y.ravel())
    m2 = MBPLS(n_components=3).fit([X], Y)
    assert np.array_equal(m0.beta_, m1.beta_) and np.array_equal(m0.beta_, m2.beta_)
    Xc = X.copy()
    MBPLS(n_components=2, copy=False).fit(X, Y)
    assert np.array_equal(X, Xc)                      # caller arrays are never modified
    for nm, o in (("1", 1), ("inf", np.inf), ("-inf", -np.inf), ("fro", "fro"), ("nuc", "nuc"), ("-1", -1), ("-2", -2)):
        mm = MBPLS(n_components=3, nipals_convergence_norm=o).fit(X, Y)
        assert util.relerr(mm.beta_, g[f"ref/norm/{nm}/beta_"]) < TOL
    for bad in (0, 3):
        with pytest.raises(ValueError):
            MBPLS(n_components=2, nipals_convergence_norm=bad).fit(X, Y)
    with pytest.raises(NameError):
        MBPLS(method="nope").fit(X, Y)
    with pytest.raises(ValueError):
        MBPLS().fit(X, Y[:-1])
    Xnan = X.copy(); Xnan[0, 0] = np.nan
    with pytest.raises(ValueError):
        MBPLS().fit(Xnan, Y)
    with pytest.raises(ValueError):
        MBPLS(standardize=False).fit(Xnan, Y)
    Xinf = X.copy(); Xinf[3, 7] = np.inf
    with pytest.raises(ValueError):
        MBPLS(method="KERNEL").fit(Xinf, Y)
    with pytest.raises(ValueError):
        m0.predict(Xnan)
    from sklearn.exceptions import NotFittedError
    with pytest.raises(NotFittedError):
        MBPLS().predict(X)
    X3, Y3 = g["q3/X"], g["q3/Y"]
    for nm, o in (("2", 2), ("1", 1), ("inf", np.inf)):
        mm = MBPLS(n_components=3, nipals_convergence_norm=o).fit([X3[:, :50], X3[:, 50:]], Y3)
        util.compare(_fit_flat(mm, [X3[:, :50], X3[:, 50:]], Y3), util.sub(g, f"ref/q3/norm{nm}"), TOL, what=nm)


def test_sklearn_plumbing(gpu_required):
    from sklearn.base import clone
    from mbpls_amd import MBPLS
    m = MBPLS(n_components=3, method="UNIPALS", calc_all=False)
    c = clone(m)
    assert c.get_params() == m.get_params()
    X, Y = orc.make_latent(60, 300, 1, r=5, seed=3)
    yh = c.fit_predict(X, Y)
    assert yh.shape == (60, 1)
    assert 0.0 < c.r2_score(X, Y) <= 1.0
    Ts, U = c.fit_transform(X, Y)
    assert Ts.shape == (60, 3) and U.shape == (60, 3)


# ------------------------------------------------------------------ medium size vs the oracle
@pytest.mark.parametrize("method,n,p,q,K,blocks", [
    ("NIPALS", 4096, 20000, 1, 5, [20000]),
    ("NIPALS", 1024, 9000, 3, 3, [3000, 2500, 3500]),
    ("UNIPALS", 4096, 18000, 1, 4, [5000, 6000, 7000]),
    ("UNIPALS", 512, 12000, 4, 3, [12000]),
    ("KERNEL", 2048, 16000, 1, 5, [6000, 10000]),
    ("KERNEL", 1000, 9000, 3, 4, [9000]),
    ("SIMPLS", 2048, 15000, 2, 4, [15000]),
    ("NIPALS", 8192, 9000, 1, 3, [4000, 5000]),
    ("NIPALS", 10000, 3000, 1, 2, [3000]),          # n > 8192: long-column fallback kernels
    ("UNIPALS", 4096, 12000, 2, 12, [12000]),       # K = 12 lazily deflated components, q = 2
    ("KERNEL", 4096, 3000, 3, 6, [1000, 2000]),     # n >= p (mbpls.py:580-692): rank-1 deflation of X^T X on the device
    ("KERNEL", 2500, 2500, 1, 4, [2500]),           # n == p takes the same branch
])
def test_medium_vs_oracle(gpu_required, monkeypatch, method, n, p, q, K, blocks):
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(n, p, q, seed=n + p)
    Xn, Yn = orc.make_latent(64, p, q, seed=1)
    cuts = np.cumsum(blocks)[:-1]
    sp = (lambda A: A) if len(blocks) == 1 else (lambda A: [np.ascontiguousarray(b) for b in np.split(A, cuts, axis=1)])
    ref = orc.fit(sp(X), Y, n_components=K, method=method, full_svd=True)
    rflat = util.model_to_flat(ref)
    rflat["predict"] = orc.predict(ref, sp(Xn))
    if method != "SIMPLS":
        rTs, rT, rU = orc.transform(ref, sp(Xn), Yn, return_block_scores=True)
        for b, blk in enumerate(rT):
            rflat[f"T_new/{b}"] = blk
    else:
        rTs, rU = orc.transform(ref, sp(Xn), Yn)
    rflat["Ts_new"], rflat["U_new"] = rTs, rU
    # one-response NIPALS: the default route (one pass per component), then the reference's two passes per component;
    # KERNEL p > n: the association matrix ("1"), then the maintained thin products ("0")
    for q1_route in (("1", "0") if ((method == "NIPALS" and q == 1) or (method == "KERNEL" and p > n)) else ("1",)):
        monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", q1_route)
        monkeypatch.setenv("MBPLS_KERNEL_GRAM", q1_route)
        mdl = MBPLS(n_components=K, method=method, full_svd=True).fit(sp(X), Y)
        if method == "KERNEL" and p > n:
            assert ("matrix cores" in mdl.fit_plan_["solver"]) == (q1_route == "1"), mdl.fit_plan_
        got = _fit_flat(mdl, sp(Xn), Yn)
        w = util.compare(got, rflat, TOL, what=f"{method} n={n} p={p} route {q1_route}")
        assert w < 1e-8, w
        if method == "NIPALS" and q == 1:
            assert ref.nipals_iterations_ == [2] * K
            # two-pass route: lazy deflation (K < 39) with re-orthogonalised sweep vectors reaches the stop rule
            # ||ts_old - ts|| <= 1e-14 (mbpls.py:841) on the second iteration of EVERY component, as the explicitly
            # deflated reference does; the default route makes the one pass that matters
            util.check_nipals_iterations(mdl, ref, K)
            assert ("fixed point" in mdl.fit_plan_["solver"]) == (q1_route == "1"), mdl.fit_plan_


@pytest.mark.parametrize("p,c1,c2", [(70, 2, 2), (1001, 3, 1), (50000, 10, 10), (300001, 20, 7), (4097, 32, 32)])
def test_small_dense_helpers(ctx, p, c1, c2):
    """cross_product / right_multiply through the engine's slot interface."""
    from mbpls_amd.engine import HipEngine
    from mbpls_amd import DeviceMatrix
    rng = np.random.RandomState(p)
    eng = HipEngine(ctx)
    eng.load_device(DeviceMatrix(ctx, 8, p), copy=False)
    A, Bm = rng.standard_normal((p, c1)), rng.standard_normal((p, c2))
    eng.put("A", A); eng.put("B", Bm)
    assert util.relerr(eng.cross(("A", 0, c1), ("B", 0, c2)), A.T.dot(Bm)) < TIGHT
    if c1 > 1:
        assert util.relerr(eng.cross(("A", 1, c1 - 1), ("B", 0, 1)), A[:, 1:].T.dot(Bm[:, :1])) < TIGHT
    M = rng.standard_normal((c1, c2))
    eng.slot("C", c2)
    eng.right_mul(("A", 0, c1), M, ("C", 0))
    assert util.relerr(eng.fetch("C"), A.dot(M)) < TIGHT
    r0, r1 = p // 3, p // 3 + max(1, p // 2)
    eng.slot("D", c1)
    eng.right_mul(("A", 0, c1), np.diag(np.arange(1, c1 + 1.0)), ("D", 0), rows=(r0, r1))
    Dh = eng.fetch("D")
    assert util.relerr(Dh[r0:r1], A[r0:r1] * np.arange(1, c1 + 1.0)) < TIGHT
    assert not Dh[:r0].any() and not Dh[r1:].any()
    eng.close()


@pytest.mark.parametrize("p,c1,c2", [(5000, 33, 32), (5000, 34, 33), (70001, 41, 40), (3000, 101, 100), (900, 70, 33)])
def test_right_multiply_in_place_beyond_one_tile(ctx, p, c1, c2):
    """C = A M with more than 32 columns runs over 32 x 32 tiles of M, the tiles along c1 accumulating into C; IN PLACE
    (C = the leading columns of A: the update S <- S [I; -r] of the maintained products) a later tile read columns an
    earlier one had overwritten -- several-response UNIPALS with q >= 33 had components 2 ... K wrong by 2-25 % (found by
    a probe with 40 and 100 responses; the randomised sweep stops at 12).  The aliased tiled product goes through the
    workspace now."""
    from mbpls_amd.engine import HipEngine
    from mbpls_amd import DeviceMatrix
    rng = np.random.RandomState(c1)
    eng = HipEngine(ctx)
    eng.load_device(DeviceMatrix(ctx, 8, p), copy=False)
    A = rng.standard_normal((p, c1))
    eng.put("A", A)
    M = np.vstack([np.eye(c2), rng.standard_normal((c1 - c2, c2))]) if c1 > c2 else rng.standard_normal((c1, c2))
    eng.right_mul(("A", 0, c1), M, ("A", 0))                       # in place over the leading c2 columns
    got = eng.fetch("A")
    assert util.relerr(got[:, :c2], A.dot(M)) < TIGHT
    assert np.array_equal(got[:, c2:], A[:, c2:])                  # the other columns are untouched
    G = rng.standard_normal((c1, c2))
    eng.put("A", A)
    eng.right_mul(("A", 0, c1), G, ("A", 0))                       # a general M, in place
    assert util.relerr(eng.fetch("A")[:, :c2], A.dot(G)) < TIGHT
    eng.close()


@pytest.mark.parametrize("q", [33, 40, 100])
def test_many_responses(gpu_required, q, kernel_route):
    """more responses than one tile of the device helpers holds (32): every method against the oracle"""
    from mbpls_amd import MBPLS
    n, p, K = 200, 5000, 3
    X, Y = orc.make_latent(n, p, q, r=6, seed=q)
    blocks = [np.ascontiguousarray(X[:, :2200]), np.ascontiguousarray(X[:, 2200:])]
    for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
        ref = orc.fit(blocks, Y, n_components=K, method=method, full_svd=True, max_iter=3000)
        m = MBPLS(n_components=K, method=method, full_svd=True).fit(blocks, Y)
        w = util.compare(util.model_to_flat(m), util.model_to_flat(ref), TOL, what=f"q={q} {method}")
        assert w < 1e-9, (method, w)
        assert util.relerr(m.predict(blocks), orc.predict(ref, blocks)) < 1e-9


def test_q1_fixed_point_option(gpu_required, monkeypatch):
    """One pass per component for one-response NIPALS (the default; MBPLS_NIPALS_Q1_FIXED_POINT=1 says so explicitly):
    same attributes as the reference's two passes."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(4096, 12000, 1, seed=21)
    blocks = [np.ascontiguousarray(X[:, :5000]), np.ascontiguousarray(X[:, 5000:])]
    ref = orc.fit(blocks, Y, n_components=4, method="NIPALS")
    monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", "1")
    mdl = MBPLS(n_components=4).fit(blocks, Y)
    assert mdl.nipals_iterations_ == [1, 1, 1, 1]
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="q1 fixed point")
    assert w < 1e-9


@pytest.mark.parametrize("method,nblocks,K,n", [("NIPALS", 1, 6, 1024), ("NIPALS", 3, 7, 1024), ("UNIPALS", 2, 12, 1024),
                                                ("NIPALS", 6, 5, 300), ("UNIPALS", 6, 6, 5000), ("NIPALS", 2, 5, 5000)])
def test_r_and_beta_column_by_column(gpu_required, monkeypatch, method, nblocks, K, n):
    """A device-resident fit builds R_ = W (P^T W)^+ and beta_ = R_ V^T (mbpls.py:988-989 / :569-570) one column per
    component while it runs (P^T W~ is upper triangular; its columns come with the lazy-deflation sums): same
    attributes as the product behind the last pass (the fallback form, forced) and as the oracle, including the deep
    components of a long fit; a rank-deficient matrix -- where the reference's pinv truncates -- takes the product.
    (n = 300 / 1024: the register-resident step kernels; six blocks and n = 5000: the general one.)"""
    from mbpls_amd import MBPLS
    p = 6000
    X, Y = orc.make_latent(n, p, 1, r=20, seed=33)
    cuts = [0] + [p * (b + 1) // nblocks for b in range(nblocks)]
    blocks = [np.ascontiguousarray(X[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    ref = orc.fit(blocks, Y, n_components=K, method=method)
    mdl = MBPLS(n_components=K, method=method).fit(blocks, Y)
    assert mdl.fit_plan_["r_and_beta"] == "column by column during the fit"
    with util.forced_routes(r_incremental=False):
        old = MBPLS(n_components=K, method=method).fit(blocks, Y)
    assert old.fit_plan_["r_and_beta"] == "one product behind the last pass"
    for got in (mdl, old):
        assert util.compare(util.model_to_flat(got), util.model_to_flat(ref), TOL, what=f"{method} R/beta") < 1e-9
    assert util.relerr(mdl.R_, old.R_) < 1e-11 and util.relerr(mdl.beta_, old.beta_) < 1e-11
    # Ts = X R with the fitted scalers: the defining property of R_
    Z = np.hstack([(b - sc.mean_) / sc.scale_ for b, sc in zip(blocks, mdl.x_scalers_)])
    assert util.relerr(Z.dot(mdl.R_), mdl.Ts_) < 1e-9
    # rank 3 < K = 5: the columns recorded for the exhausted components are not trusted
    rng = np.random.RandomState(5)
    Xr = rng.randn(256, 3).dot(rng.randn(3, 2000))
    yr = Xr[:, :1] + 0.1 * rng.randn(256, 1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        low = MBPLS(n_components=5, method=method, standardize=False).fit([Xr], yr)
    assert low.fit_plan_["r_and_beta"] == "one product behind the last pass"


@pytest.mark.parametrize("q", [2, 4])
def test_row_subset_fit_and_cross_val_predict(gpu_required, q):
    """Fits on a ROW SUBSET of a resident matrix == the oracle on the sliced arrays; cross_val_predict on the
    device == a loop of oracle fits (what sklearn's cross_val_predict does with the reference).  Two and four
    responses: the two ways the several-response solvers set up their maintained products."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.model_selection import cross_val_predict
    n, p = 120, 900
    X, Y = orc.make_latent(n, p, q, r=6, seed=4)
    blocks = [np.ascontiguousarray(X[:, :400]), np.ascontiguousarray(X[:, 400:])]
    ctx = mbpls_amd.get_context()
    Xd = DeviceMatrix(ctx, n, p, [0, 400, p])
    Xd.upload(X)
    rng = np.random.RandomState(0)
    mask = rng.rand(n) > 0.25
    for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
        for std in (True, False):
            ref = orc.fit([b[mask] for b in blocks], Y[mask], n_components=3, method=method, standardize=std,
                          full_svd=True)
            mdl = MBPLS(n_components=3, method=method, standardize=std)._fit(Xd, Y, mask)
            w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"subset/{method}/{std}")
            assert w < 1e-9, (method, std, w)
            assert util.relerr(mdl.predict(Xd)[~mask], orc.predict(ref, [b[~mask] for b in blocks])) < 1e-9
    assert np.array_equal(Xd.download(), X)          # the resident matrix is untouched by all those fits
    # k-fold and leave-one-out style CV (5 folds + an explicit split list)
    for cv in (5, [(np.arange(n)[np.arange(n) % 3 != f], np.arange(n)[np.arange(n) % 3 == f]) for f in range(3)]):
        got = cross_val_predict(MBPLS(n_components=2), Xd, Y, cv=cv)
        from sklearn.model_selection import check_cv
        splits = list(check_cv(cv, Y, classifier=False).split(np.zeros((n, 1)), Y)) if isinstance(cv, int) else cv
        want = np.empty_like(Y)
        for tr, te in splits:
            r = orc.fit([b[tr] for b in blocks], Y[tr], n_components=2)
            want[te] = orc.predict(r, [b[te] for b in blocks])
        assert util.relerr(got, want) < 1e-9
    # host input is uploaded once
    got2 = cross_val_predict(MBPLS(n_components=2), blocks, Y, cv=5)
    assert util.relerr(got2, cross_val_predict(MBPLS(n_components=2), Xd, Y, cv=5)) < 1e-12
    Xd.close()


@pytest.mark.parametrize("blocks,std,cv", [(1, True, "loo"), (2, True, 7), (3, False, 5), (2, True, "thirds")])
def test_cross_val_predict_batched_folds(gpu_required, q1_route, blocks, std, cv):
    """One-response NIPALS (one sweep per component and fold, or the reference's two: q1_route): up to 256 folds per pass
    over X (cv_batch) == the fold-by-fold loop == a loop of oracle fits,
    for leave-one-out, k-fold and an explicit split list, one and several blocks, with and without standardisation; a
    constant column and a column that is constant on one fold's training rows only included."""
    from mbpls_amd import MBPLS, cv_batch
    from mbpls_amd.model_selection import cross_val_predict
    n, p, K = 60, 700, 4
    X, Y = orc.make_latent(n, p, 1, r=6, seed=21)
    X[:, 5] = 3.0                                    # constant everywhere
    X[:, 9] = 1.0; X[0, 9] = 2.0                     # constant on the training rows of the fold that holds out row 0
    cuts = np.linspace(0, p, blocks + 1).astype(int)
    Xs = X if blocks == 1 else [np.ascontiguousarray(X[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    if cv == "loo":
        cv_arg, splits = n, [(np.delete(np.arange(n), i), np.array([i])) for i in range(n)]
    elif cv == "thirds":
        splits = [(np.arange(n)[np.arange(n) % 3 != f], np.arange(n)[np.arange(n) % 3 == f]) for f in range(3)]
        cv_arg = splits
    else:
        from sklearn.model_selection import KFold
        cv_arg, splits = cv, list(KFold(cv).split(X))
    est = MBPLS(n_components=K, standardize=std)
    assert cv_batch.eligible(est, 1, n, p, len(splits))
    calls = []
    real = cv_batch.predict_folds
    cv_batch.predict_folds = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = cross_val_predict(est, Xs, Y, cv=cv_arg)
            loop = cross_val_predict(est, Xs, Y, cv=cv_arg, batch_folds=False)
    finally:
        cv_batch.predict_folds = real
    assert calls == [1]                              # the batched path ran (once), the second call looped
    sp = (lambda A, idx: A[idx]) if blocks == 1 else (lambda A, idx: [b[idx] for b in A])
    want = np.empty_like(Y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tr, te in splits:
            r = orc.fit(sp(Xs, tr), Y[tr], n_components=K, standardize=std)
            want[te] = orc.predict(r, sp(Xs, te))
    assert util.relerr(got, want) < 1e-9
    assert util.relerr(got, loop) < 1e-9


@pytest.mark.parametrize("method,q,blocks,cv", [("UNIPALS", 1, 2, "loo"), ("KERNEL", 1, 1, 6), ("SIMPLS", 1, 3, 5),
                                                ("NIPALS", 3, 2, "loo"), ("NIPALS", 2, 1, 4),
                                                ("UNIPALS", 3, 2, "loo"), ("KERNEL", 2, 3, 6)])
def test_cross_val_predict_batched_folds_other_routes(gpu_required, method, q, blocks, cv):
    """Round 4: one response with EVERY method (one model, the batched passes without the confirming iteration) and
    several-response NIPALS (every fold's inner loop advances together) == a loop of oracle fits of that method."""
    from mbpls_amd import MBPLS, cv_batch
    from mbpls_amd.model_selection import cross_val_predict
    n, p, K = 48, 500, 3
    X, Y = orc.make_latent(n, p, q, r=6, seed=33 + q)
    X[:, 7] = -2.0                                   # a constant column
    cuts = np.linspace(0, p, blocks + 1).astype(int)
    Xs = X if blocks == 1 else [np.ascontiguousarray(X[:, a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    if cv == "loo":
        cv_arg, splits = n, [(np.delete(np.arange(n), i), np.array([i])) for i in range(n)]
    else:
        from sklearn.model_selection import KFold
        cv_arg, splits = cv, list(KFold(cv).split(X))
    est = MBPLS(n_components=K, method=method, full_svd=True)
    assert cv_batch.eligible(est, q, n, p, len(splits))
    calls = []
    real = cv_batch.predict_folds
    cv_batch.predict_folds = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            got = cross_val_predict(est, Xs, Y, cv=cv_arg)
    finally:
        cv_batch.predict_folds = real
    assert calls == [1] and got.shape == Y.shape
    sp = (lambda A, idx: A[idx]) if blocks == 1 else (lambda A, idx: [b[idx] for b in A])
    want = np.empty_like(Y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for tr, te in splits:
            r = orc.fit(sp(Xs, tr), Y[tr], n_components=K, method=method, full_svd=True)
            want[te] = orc.predict(r, sp(Xs, te))
    assert util.relerr(got, want) < 1e-9, util.relerr(got, want)


def test_cross_val_predict_batched_folds_validate_like_fit(gpu_required):
    """advisor (round 3): the batched route skipped the validation of fit() when standardize=False -- a NaN in X gave
    NaN predictions instead of fit()'s ValueError.  The statistics pass now always runs."""
    from mbpls_amd import MBPLS
    from mbpls_amd.model_selection import cross_val_predict
    X, Y = orc.make_latent(40, 300, 1, r=4, seed=2)
    X[3, 17] = np.nan
    for std in (True, False):
        with pytest.raises(ValueError):
            cross_val_predict(MBPLS(n_components=2, standardize=std), X, Y, cv=8)
    # ... and a splitter whose test sets are not a partition of the samples is refused as sklearn refuses it
    from sklearn.model_selection import GroupKFold, KFold, ShuffleSplit
    X, Y = orc.make_latent(40, 300, 1, r=4, seed=2)
    with pytest.raises(ValueError, match="only works for partitions"):
        cross_val_predict(MBPLS(n_components=2), X, Y, cv=ShuffleSplit(5, test_size=0.2, random_state=0))
    with pytest.raises(ValueError, match="only works for partitions"):
        cross_val_predict(MBPLS(n_components=2), X, Y, cv=list(KFold(4).split(X))[:3])
    groups = np.repeat(np.arange(8), 5)
    got = cross_val_predict(MBPLS(n_components=2), X, Y, cv=GroupKFold(4), groups=groups)
    want = np.empty_like(Y)
    for tr, te in GroupKFold(4).split(X, Y, groups):
        want[te] = orc.predict(orc.fit(X[tr], Y[tr], n_components=2), X[te])
    assert util.relerr(got, want) < 1e-9


@pytest.mark.parametrize("n,p,q,K", [(5, 1, 1, 1), (3, 2, 1, 1), (7, 3, 2, 2), (64, 1, 1, 1), (2, 5, 1, 1), (6, 4, 1, 3),
                                     (9, 300, 3, 3), (513, 2, 1, 2)])
def test_tiny_and_ragged_shapes(gpu_required, n, p, q, K):
    """Smallest inputs the reference accepts: one feature, two samples, fewer columns than compute units,
    odd sample counts (padded columns) -- every method against the oracle."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(n, p, q, r=min(3, p), seed=n + p)
    for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ref = orc.fit(X, Y, n_components=K, method=method, full_svd=True)
            mdl = MBPLS(n_components=K, method=method, full_svd=True).fit(X, Y)
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} {n}x{p}")
        assert w < 1e-9, (method, w)
        assert util.relerr(mdl.predict(X), orc.predict(ref, X)) < 1e-9


@pytest.mark.parametrize("mode", ["lazy", "explicit"])
@pytest.mark.parametrize("method,q", [("NIPALS", 1), ("NIPALS", 3), ("UNIPALS", 2)])
def test_both_deflation_modes(gpu_required, monkeypatch, mode, method, q):
    """Implicit (X never rewritten) and explicit (in-place) deflation give the reference's numbers; the default
    picks lazy for one response and explicit for several (many NIPALS iterations per component)."""
    from mbpls_amd import MBPLS
    monkeypatch.setenv("MBPLS_DEFLATION", mode)
    X, Y = orc.make_latent(700, 5000, q, seed=21 + q)
    Xs = [np.ascontiguousarray(X[:, :2000]), np.ascontiguousarray(X[:, 2000:])]
    ref = _cached_oracle(("deflation modes", method, q), lambda: orc.fit(Xs, Y, n_components=5, method=method))
    mdl = MBPLS(n_components=5, method=method).fit(Xs, Y)
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} q={q} {mode}")
    assert w < 1e-8, w


@pytest.mark.parametrize("offset,n,p", [(0.0, 1000, 6000), (3.0, 1000, 6000), (50.0, 1000, 6000), (1.0e3, 1000, 6000),
                                        (1.0e3, 300, 40), (5.0e3, 64, 7), (1.0e6, 1000, 6000)])
@pytest.mark.parametrize("resident", ["1", "0"])
def test_implicit_standardisation_of_offset_data(gpu_required, monkeypatch, q1_route, offset, n, p, resident):
    """One response + lazy deflation: the fit never writes X (statistics-only first pass, raw-space passes after).
    The statistics pass shifts every element by the column's first sample and reports max |mean|/scale; above
    HipEngine.CENTRE_FROM_OFFSET the later passes subtract the column mean in registers before every dot product and
    accumulation, so column offsets of 10^3..10^6 scale units neither cost accuracy nor keep ||ts_old - ts|| from
    reaching 1e-14: the iteration counts equal the reference's ([2]*K), with the device-resident queue (which finds
    out on the device and is queued again with centring passes) and with the host-driven loop.  The storing
    standardise pass remains behind MBPLS_IMPLICIT_STD=0.  The caller's resident matrix is untouched."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.engine import HipEngine
    monkeypatch.setenv("MBPLS_RESIDENT", resident)
    K = 4
    X, Y = orc.make_latent(n, p, 1, seed=77)
    X = X + offset * (1.0 + np.arange(X.shape[1]) % 3)       # column offsets 1x..3x
    ref = orc.fit(X, Y, n_components=K)
    seen, cen = [], []
    real = HipEngine._prepare_implicit
    real_close = HipEngine.close

    def spy(self, *a, **k):
        seen.append(True)
        return real(self, *a, **k)

    def spy_close(self):
        cen.append(self._cen)
        return real_close(self)

    monkeypatch.setattr(HipEngine, "_prepare_implicit", spy)
    monkeypatch.setattr(HipEngine, "close", spy_close)
    ctx = mbpls_amd.get_context()
    Xd = DeviceMatrix(ctx, *X.shape)
    Xd.upload(X)
    mdl = MBPLS(n_components=K).fit(Xd, Y)
    centred = bool(np.max(np.abs(X.mean(axis=0)) / X.std(axis=0)) > HipEngine.CENTRE_FROM_OFFSET)
    assert centred == (offset >= 50.0) or offset == 3.0      # both flavours of the passes are exercised
    assert cen == [centred]
    assert len(seen) == (2 if centred and resident == "1" else 1)      # queued twice when the device finds offset data
    util.check_nipals_iterations(mdl, ref, K)
    w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"offset {offset}")
    # the input itself is only known to offset * eps scale units: that, not the arithmetic, bounds the agreement
    assert w < max(1e-9, 1e3 * offset * np.finfo(float).eps), w
    assert np.array_equal(Xd.download(), X)                  # copy=True: the resident matrix is never written
    assert util.relerr(np.concatenate([s.mean_ for s in mdl.x_scalers_]), X.mean(axis=0)) < 1e-12
    assert util.relerr(np.concatenate([s.scale_ for s in mdl.x_scalers_]), X.std(axis=0)) < max(1e-9, 1e2 * offset * 2.3e-16)
    monkeypatch.setenv("MBPLS_IMPLICIT_STD", "0")
    seen.clear()
    m2 = MBPLS(n_components=K).fit(Xd, Y)
    assert seen == []
    util.check_nipals_iterations(m2, ref, K)
    assert util.compare(util.model_to_flat(m2), util.model_to_flat(ref), TOL, what="stored") < max(1e-8, 1e3 * offset * 2.3e-16)
    Xd.close()


@pytest.mark.parametrize("method", ["NIPALS", "UNIPALS", "KERNEL", "SIMPLS"])
@pytest.mark.parametrize("q,blocks", [(1, [9000]), (3, [800, 1000, 1200])])
def test_medium_vs_faithful_oracle(gpu_required, monkeypatch, method, q, blocks):
    """The HIP path against the oracle's FAITHFUL mode -- the reference's own numpy op sequence (n x p outer products for
    the explained variance :933/:946, n x n SVDs :489-493 / :712-715, D A D deflation :722-725, per-iteration sweeps),
    not the lean restatement the GPU shares its formulas with -- at a size where that is still quick."""
    from mbpls_amd import MBPLS
    n, p, K = 600, sum(blocks), 4
    X, Y = orc.make_latent(n, p, q, seed=200 + q)
    cuts = np.cumsum(blocks)[:-1]
    Xs = X if len(blocks) == 1 else [np.ascontiguousarray(b) for b in np.split(X, cuts, axis=1)]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.fit(Xs, Y, n_components=K, method=method, full_svd=True, faithful=True)
        for kg in (("1", "0") if method == "KERNEL" else ("1",)):      # (KERNEL: both of its p > n routes)
            monkeypatch.setenv("MBPLS_KERNEL_GRAM", kg)
            mdl = MBPLS(n_components=K, method=method, full_svd=True).fit(Xs, Y)
            w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"faithful {method} q={q} {kg}")
            assert w < 1e-8, w
            assert util.relerr(mdl.predict(Xs), orc.predict(ref, Xs)) < 1e-8
    if method == "NIPALS":
        assert all(abs(a - b) <= 2 for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_))


@pytest.mark.parametrize("method,K,q,n,p", [("NIPALS", 40, 1, 1200, 900), ("NIPALS", 38, 1, 1200, 900),
                                            ("NIPALS", 33, 2, 600, 500), ("UNIPALS", 40, 1, 300, 2400),
                                            ("UNIPALS", 35, 3, 300, 2400), ("KERNEL", 40, 2, 300, 2400),
                                            ("SIMPLS", 40, 1, 300, 2400)])
def test_many_components(gpu_required, monkeypatch, method, K, q, n, p):
    """More components than one 32-column tile of the device helpers (cross products / right-multiplications are tiled)
    and than the lazy-deflation correction registers hold (K >= 39 takes the stored, explicitly deflated path): any K
    works, as in the reference.  (NIPALS on tall data: with p > n the response is fitted exactly after ~30 components
    and the stop rule of mbpls.py:841 is then out of reach for the reference as well.)"""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(n, p, q, r=60, seed=5 + K, decay=0.97, noise_y=1.0)
    h = p // 2 - 50
    Xs = [np.ascontiguousarray(X[:, :h]), np.ascontiguousarray(X[:, h:])]
    ref = orc.fit(Xs, Y, n_components=K, method=method)
    # one-response NIPALS: the default route (one pass per component), then the reference's two passes per component;
    # KERNEL: the association matrix, then the maintained thin products
    for q1_route in (("1", "0") if ((method == "NIPALS" and q == 1) or method == "KERNEL") else ("1",)):
        monkeypatch.setenv("MBPLS_NIPALS_Q1_FIXED_POINT", q1_route)
        monkeypatch.setenv("MBPLS_KERNEL_GRAM", q1_route)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            mdl = MBPLS(n_components=K, method=method).fit(Xs, Y)
        w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} K={K} q={q} route {q1_route}")
        assert w < 1e-8, w
        if method == "NIPALS" and q == 1:
            assert ref.nipals_iterations_ == [2] * K
            # two-pass route: lazy deflation (K < 39) with re-orthogonalised sweep vectors reaches the stop rule
            # ||ts_old - ts|| <= 1e-14 (mbpls.py:841) on the second iteration of EVERY component, as the explicitly
            # deflated reference does; the default route makes the one pass that matters
            util.check_nipals_iterations(mdl, ref, K)


@pytest.mark.parametrize("n,p,K", [(512, 8192, 30), (1024, 16384, 34)])
def test_exhausted_response_iterations_follow_the_oracle(gpu_required, two_pass_nipals, n, p, K):
    """Deep components of a response that p >> n columns fit exactly: the REFERENCE algorithm (oracle, explicit
    deflation) meets ||ts_old - ts|| <= 1e-14 on the second iteration for the first 24 / 27 components and then no
    longer terminates (iteration cap).  The lazily deflated fit -- sweep vectors re-orthogonalised against the earlier
    scores -- takes exactly the oracle's iterations for every component of that converging stretch (device-resident
    queue and resumed host loop), gives up on the others once the difference stagnates (solvers.q1_noise_floor) with a warning, and
    says which in nipals_converged_."""
    from mbpls_amd import MBPLS, solvers
    X, Y = orc.make_latent(n, p, 1, seed=3)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = orc.fit(X, Y, n_components=K, method="NIPALS", faithful=False, max_iter=12)
    good = next(i for i, it in enumerate(ref.nipals_iterations_) if it != 2)
    assert 20 <= good < K, ref.nipals_iterations_
    good -= 2       # (the last components before the oracle gives up sit at the noise floor themselves: either may take a pass more)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        mdl = MBPLS(n_components=K).fit(X, Y)
    assert list(mdl.nipals_iterations_[:good]) == [2] * good, (mdl.nipals_iterations_, ref.nipals_iterations_)
    assert all(mdl.nipals_converged_[:good])
    assert all(it <= solvers.NIPALS_Q1_NOISE_FLOOR_MAX for it in mdl.nipals_iterations_)
    stopped = [k for k, c in enumerate(mdl.nipals_converged_) if not c]
    assert all(k >= good for k in stopped)
    assert len([w for w in rec if "stays above max_tol" in str(w.message)]) == len(stopped)
    # the converging stretch agrees with the oracle to the usual tolerance
    s = np.sign(np.sum(ref.Ts_[:, :good] * mdl.Ts_[:, :good], axis=0))
    assert util.relerr(mdl.Ts_[:, :good] * s, ref.Ts_[:, :good]) < 1e-8
    assert util.relerr(mdl.P_[0][:, :good] * s, ref.P_[0][:, :good]) < 1e-8


@pytest.mark.parametrize("blocks,standardize,norm,implicit", [(1, True, 2, "1"), (3, True, 2, "1"), (3, False, 1, "1"),
                                                              (2, True, np.inf, "0"), (1, True, -np.inf, "1"),
                                                              (16, True, "fro", "1")])
def test_device_resident_nipals(gpu_required, two_pass_nipals, monkeypatch, blocks, standardize, norm, implicit):
    """One response: the whole NIPALS fit is queued on the device (csrc/resident.hip does the n x B algebra and the
    stop test between the passes; no host round trip per pass).  Same numbers as the host-driven loop
    (MBPLS_RESIDENT=0) and as the oracle, for every variant of the passes (implicit / stored standardisation, no
    standardisation, several blocks, every norm of the stop rule)."""
    from mbpls_amd import MBPLS
    from mbpls_amd.engine import HipEngine
    monkeypatch.setenv("MBPLS_IMPLICIT_STD", implicit)
    K = 6
    X, Y = orc.make_latent(900, 4800, 1, seed=31 + blocks)
    Xs = [np.ascontiguousarray(b) for b in np.array_split(X, blocks, axis=1)] if blocks > 1 else X
    ref = orc.fit(Xs, Y, n_components=K, standardize=standardize, nipals_convergence_norm=norm)
    calls = []
    real = HipEngine.nipals_resident

    def spy(self, *a, **k):
        out = real(self, *a, **k)
        calls.append(out["k_done"])
        return out

    monkeypatch.setattr(HipEngine, "nipals_resident", spy)
    mdl = MBPLS(n_components=K, standardize=standardize, nipals_convergence_norm=norm).fit(Xs, Y)
    assert calls == [K]                                      # the whole fit ran on the device queue
    assert mdl.nipals_iterations_ == ref.nipals_iterations_ == [2] * K
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="resident") < 1e-9
    monkeypatch.setenv("MBPLS_RESIDENT", "0")
    calls.clear()
    m0 = MBPLS(n_components=K, standardize=standardize, nipals_convergence_norm=norm).fit(Xs, Y)
    assert calls == []
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(m0), TOL, what="resident vs host loop") < 1e-11
    assert util.relerr(mdl.predict(Xs), orc.predict(ref, Xs)) < 1e-9


@pytest.mark.parametrize("blocks,standardize,offset", [(1, True, 0.0), (3, True, 0.0), (2, False, 0.0), (3, True, 200.0)])
def test_device_resident_unipals(gpu_required, monkeypatch, blocks, standardize, offset):
    """UNIPALS, p > n, one response: a component is one pass, the n x B algebra between the passes runs in the step
    kernel (mode 1) and the whole fit is one device queue -- same numbers as the host-driven loop and as the oracle,
    also for offset data (queued again with the centring passes)."""
    from mbpls_amd import MBPLS
    from mbpls_amd.engine import HipEngine
    K = 5
    X, Y = orc.make_latent(800, 4200, 1, seed=41 + blocks)
    X = X + offset
    Xs = [np.ascontiguousarray(b) for b in np.array_split(X, blocks, axis=1)] if blocks > 1 else X
    ref = orc.fit(Xs, Y, n_components=K, method="UNIPALS", standardize=standardize)
    calls = []
    real = HipEngine.unipals_resident

    def spy(self, *a, **k):
        out = real(self, *a, **k)
        calls.append(out.get("reason", out["k_done"]))
        return out

    monkeypatch.setattr(HipEngine, "unipals_resident", spy)
    import mbpls_amd
    mbpls_amd.get_context()._offset_hints.clear()          # (what earlier fits of this shape found out: see below)
    mdl = MBPLS(n_components=K, method="UNIPALS", standardize=standardize).fit(Xs, Y)
    assert calls == (["offset", K] if offset and standardize else [K])
    tol = max(1e-9, 1e3 * offset * np.finfo(float).eps)
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="resident unipals") < tol
    # the next fit of host data of this shape starts with the flavour this one needed (engine.HipEngine._offset_hint):
    # one queue, the same numbers bit for bit -- and goes back to plain passes once the data no longer ask for centring
    calls.clear()
    again = MBPLS(n_components=K, method="UNIPALS", standardize=standardize).fit(Xs, Y)
    assert calls == [K]
    a, b = util.model_to_flat(again), util.model_to_flat(mdl)
    assert all(np.array_equal(a[k], b[k]) for k in b)
    if offset and standardize:
        Xc = X - X.mean(axis=0)
        Xcs = [np.ascontiguousarray(c) for c in np.array_split(Xc, blocks, axis=1)] if blocks > 1 else Xc
        refc = orc.fit(Xcs, Y, n_components=K, method="UNIPALS", standardize=standardize)
        for expect_centring in (True, False):
            calls.clear()
            mc = MBPLS(n_components=K, method="UNIPALS", standardize=standardize).fit(Xcs, Y)
            assert calls == [K] and mc.fit_plan_["centred_in_registers"] is expect_centring, (calls, mc.fit_plan_)
            assert util.compare(util.model_to_flat(mc), util.model_to_flat(refc), TOL, what="hint, centred data") < 1e-9
    calls.clear()
    monkeypatch.setenv("MBPLS_RESIDENT", "0")
    calls.clear()
    m0 = MBPLS(n_components=K, method="UNIPALS", standardize=standardize).fit(Xs, Y)
    assert calls == []
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(m0), TOL, what="resident vs host loop") < max(1e-11, tol)
    assert util.relerr(mdl.predict(Xs), orc.predict(ref, Xs)) < tol


def test_offset_hint_of_a_resident_matrix(gpu_required, monkeypatch):
    """Raw spectra sit many standard deviations from zero: a device-resident queue finds that out in its statistics pass,
    stops, and the fit is queued again with centring passes.  The next fit of THAT matrix starts with them (the measure
    is kept on the DeviceMatrix object); another matrix of the same shape learns it for itself; a one-pass-late return
    to plain passes when the content no longer asks for centring.  Same numbers either way."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.engine import HipEngine
    ctx = mbpls_amd.get_context()
    n, p, K = 600, 5000, 3
    X, Y = orc.make_latent(n, p, 1, seed=5)
    Xo = X + 40.0
    ref = orc.fit(Xo, Y, n_components=K)
    calls = []

    def spied(real):
        def spy(self, *a, **k):
            out = real(self, *a, **k)
            calls.append(out.get("reason", out["k_done"]))
            return out
        return spy

    for name in ("nipals_resident", "unipals_resident"):      # (one-response NIPALS queues its one-pass form by default)
        monkeypatch.setattr(HipEngine, name, spied(getattr(HipEngine, name)))
    D = DeviceMatrix(ctx, n, p)
    D.upload(Xo, 0)
    first = MBPLS(n_components=K).fit(D, Y)
    assert calls == ["offset", K], calls
    calls.clear()
    second = MBPLS(n_components=K).fit(D, Y)
    assert calls == [K] and second.fit_plan_["centred_in_registers"] is True
    a, b = util.model_to_flat(first), util.model_to_flat(second)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert util.compare(b, util.model_to_flat(ref), TOL, what="offset hint") < 1e-9
    E = DeviceMatrix(ctx, n, p)                      # another matrix of this shape: no hint of its own yet
    E.upload(Xo, 0)
    calls.clear()
    MBPLS(n_components=K).fit(E, Y)
    assert calls == ["offset", K], calls
    E.close()
    D.upload(X - X.mean(axis=0), 0)                  # new content, centred: one fit later the plain passes are back
    refc = orc.fit(X - X.mean(axis=0), Y, n_components=K)
    for expect in (True, False):
        calls.clear()
        m = MBPLS(n_components=K).fit(D, Y)
        assert calls == [K] and m.fit_plan_["centred_in_registers"] is expect
        assert util.compare(util.model_to_flat(m), util.model_to_flat(refc), TOL, what="offset hint, centred content") < 1e-9
    D.close()


def test_unknown_big_matrix_asks_its_statistics_pass(gpu_required, monkeypatch):
    """From 1 GiB on a pass takes longer than the queue takes to build: a matrix nothing is known about waits for its
    statistics pass and queues the right flavour at once -- no queue thrown away, also on the first fit."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.engine import HipEngine
    ctx = mbpls_amd.get_context()
    n, p, K = 4096, 36000, 2                          # 1.18 GB
    X, Y = orc.make_latent(n, p, 1, seed=9)
    X += 25.0
    ref = orc.fit(X, Y, n_components=K)
    calls = []

    def spied(real):
        def spy(self, *a, **k):
            out = real(self, *a, **k)
            calls.append(out.get("reason", out["k_done"]))
            return out
        return spy

    for name in ("nipals_resident", "unipals_resident"):
        monkeypatch.setattr(HipEngine, name, spied(getattr(HipEngine, name)))
    D = DeviceMatrix(ctx, n, p)
    D.upload(X, 0)
    m = MBPLS(n_components=K).fit(D, Y)
    assert calls == [K] and m.fit_plan_["centred_in_registers"] is True, (calls, m.fit_plan_)
    assert util.compare(util.model_to_flat(m), util.model_to_flat(ref), TOL, what="asked its statistics pass") < 1e-9
    monkeypatch.setattr(HipEngine, "OFFSET_ASK_BYTES", 1 << 40)
    D._offset_hint = None                             # (nothing known again, and too small to ask: the speculative queue)
    calls.clear()
    m2 = MBPLS(n_components=K).fit(D, Y)
    assert calls == ["offset", K], calls
    a, b = util.model_to_flat(m), util.model_to_flat(m2)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    D.close()


@pytest.mark.parametrize("resume_at", [0, 3])
def test_device_resident_nipals_resumes_on_the_host(gpu_required, two_pass_nipals, monkeypatch, resume_at):
    """The device queue is speculative (two iterations per component).  When a component does not meet the stop rule
    on its second iteration the queued launches behind it turn into no-ops and the host-driven loop resumes at that
    component: forced here at component 0 (impossible tolerance on the device) and emulated at component 3."""
    from mbpls_amd import MBPLS
    from mbpls_amd.engine import HipEngine
    K = 6
    X, Y = orc.make_latent(700, 3000, 1, seed=8)
    Xs = [np.ascontiguousarray(X[:, :1300]), np.ascontiguousarray(X[:, 1300:])]
    ref = orc.fit(Xs, Y, n_components=K)
    real = HipEngine.nipals_resident
    seen = []

    def wrapped(self, Y_, K_, kind, tol, *a):
        if resume_at == 0:
            out = real(self, Y_, K_, kind, -1.0, *a)         # ||ts_old - ts|| <= -1 never holds
        else:
            out = real(self, Y_, K_, kind, tol, *a)
            assert out["k_done"] == K_
            out["k_done"] = resume_at                        # pretend component 3 needed a third iteration
            cols = [np.ascontiguousarray(out["Ts"][:, i]) for i in range(resume_at)]
            self._lazy_ts = [c - c.mean() for c in cols] if self._istd else cols
        seen.append(out["k_done"])
        return out

    monkeypatch.setattr(HipEngine, "nipals_resident", wrapped)
    mdl = MBPLS(n_components=K).fit(Xs, Y)
    assert seen == [resume_at]
    assert mdl.nipals_iterations_ == [2] * K
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="resumed") < 1e-9


@pytest.mark.parametrize("offset,standardize,blocks", [(0.0, True, 1), (300.0, True, 3), (0.0, False, 2)])
def test_kernel_method_never_writes_x(gpu_required, monkeypatch, offset, standardize, blocks):
    """KERNEL, p > n: statistics-only preparation; the Gram (MFMA SYRK), X^T [U Ts] and the block-score products
    standardise their operand in registers, so the caller's resident matrix is used as it is (no working copy, no
    read+write standardise pass) -- same numbers as the stored path (MBPLS_IMPLICIT_STD=0) and as the oracle, also for
    badly offset columns."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.engine import HipEngine
    X, Y = orc.make_latent(640, 5001, 2, seed=3)
    X = X + offset * (1.0 + np.arange(X.shape[1]) % 4)
    X[:, 17] = 2.5                                           # a constant column: scale 1, no weight
    offs = [0, 5001] if blocks == 1 else ([0, 2000, 5001] if blocks == 2 else [0, 1000, 3001, 5001])
    Xs = X if blocks == 1 else [np.ascontiguousarray(X[:, a:b]) for a, b in zip(offs[:-1], offs[1:])]
    ref = orc.fit(Xs, Y, n_components=4, method="KERNEL", standardize=standardize)
    used = []
    real = HipEngine.prepare_stats

    def spy(self, *a, **k):
        used.append(True)
        return real(self, *a, **k)

    monkeypatch.setattr(HipEngine, "prepare_stats", spy)
    monkeypatch.setenv("MBPLS_KERNEL_IMPLICIT", "1")         # (by default only when a working copy would not fit)
    monkeypatch.setenv("MBPLS_KERNEL_GRAM", "1")             # (the route on maintained products: the test below)
    ctx = mbpls_amd.get_context()
    Xd = DeviceMatrix(ctx, X.shape[0], X.shape[1], offs)
    Xd.upload(X)
    mdl = MBPLS(n_components=4, method="KERNEL", standardize=standardize).fit(Xd, Y)
    assert used == [True]
    assert np.array_equal(Xd.download(), X)                  # copy=True and nothing was written
    tol = max(1e-9, 1e3 * offset * np.finfo(float).eps)
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="kernel implicit") < tol
    if standardize:
        assert mdl.x_scalers_[0].scale_[17] == 1.0 and mdl.x_scalers_[0].var_[17] == 0.0
        assert util.relerr(np.concatenate([s.var_ for s in mdl.x_scalers_]), X.var(axis=0)) < max(1e-9, 1e2 * offset * 2.3e-16)
    monkeypatch.setenv("MBPLS_IMPLICIT_STD", "0")
    used.clear()
    m2 = MBPLS(n_components=4, method="KERNEL", standardize=standardize).fit(Xd, Y)
    assert used == []
    assert util.compare(util.model_to_flat(m2), util.model_to_flat(mdl), TOL, what="stored vs implicit") < tol
    assert util.relerr(mdl.predict(Xs), orc.predict(ref, Xs)) < tol
    Xd.close()


@pytest.mark.parametrize("offset,standardize,blocks,q,n", [(0.0, True, 1, 1, 640), (300.0, True, 3, 2, 640),
                                                           (0.0, False, 2, 4, 300), (0.0, True, 2, 3, 9000)])
def test_kernel_method_on_maintained_products(gpu_required, monkeypatch, offset, standardize, blocks, q, n):
    """KERNEL, p > n, WITHOUT the association matrix (solvers._kernel_scores_on_products): S_k = AS_X,k AS_Y,k has rank
    <= q, its leading left singular vector comes from the maintained n x q product P X (X^T (P Y)), the weights X^T u
    from the maintained p x q product -- one read-only pass over X per component, the caller's resident matrix is used
    as it is.  Same numbers as the oracle (mbpls.py:704-805) and as the matrix-core route, also for offset columns, for
    long columns (n = 9000: clusters of workgroups) and with calc_all=False."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    p = 5001 if n < 8192 else 9500
    X, Y = orc.make_latent(n, p, q, seed=3 + q)
    X = X + offset * (1.0 + np.arange(p) % 4)
    X[:, 17] = 2.5                                           # a constant column: scale 1, no weight
    offs = [0, p] if blocks == 1 else ([0, 2000, p] if blocks == 2 else [0, 1000, 3001, p])
    Xs = X if blocks == 1 else [np.ascontiguousarray(X[:, a:b]) for a, b in zip(offs[:-1], offs[1:])]
    K = 5
    ref = orc.fit(Xs, Y, n_components=K, method="KERNEL", standardize=standardize, full_svd=True)
    monkeypatch.setenv("MBPLS_KERNEL_GRAM", "0")
    ctx = mbpls_amd.get_context()
    Xd = DeviceMatrix(ctx, n, p, offs)
    Xd.upload(X)
    mdl = MBPLS(n_components=K, method="KERNEL", standardize=standardize).fit(Xd, Y)
    assert "maintained thin products" in mdl.fit_plan_["solver"], mdl.fit_plan_
    assert "never written" in mdl.fit_plan_["standardisation"], mdl.fit_plan_
    assert np.array_equal(Xd.download(), X)                  # copy=True and nothing was written
    tol = max(1e-9, 1e3 * offset * np.finfo(float).eps)
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what="kernel on products") < tol
    assert util.relerr(mdl.predict(Xs), orc.predict(ref, Xs)) < tol
    m2 = MBPLS(n_components=K, method="KERNEL", standardize=standardize, calc_all=False).fit(Xd, Y)
    assert m2.A_.shape == (blocks, 0) and all(w.shape[1] == 0 for w in m2.W_)
    for a in ("Ts_", "U_", "V_", "R_", "beta_", "W_concat_"):
        assert np.array_equal(getattr(m2, a), getattr(mdl, a)), a
    monkeypatch.setenv("MBPLS_KERNEL_GRAM", "1")
    m3 = MBPLS(n_components=K, method="KERNEL", standardize=standardize).fit(Xd, Y)
    assert "matrix cores" in m3.fit_plan_["solver"], m3.fit_plan_
    assert util.compare(util.model_to_flat(m3), util.model_to_flat(mdl), TOL, what="gram vs products") < tol
    Xd.close()


@pytest.mark.parametrize("method,offset,standardize,n", [
    ("NIPALS", 0.0, True, 640), ("NIPALS", 300.0, True, 200), ("NIPALS", 0.0, False, 640), ("NIPALS", 0.0, True, 9000),
    ("UNIPALS", 0.0, True, 640), ("UNIPALS", 300.0, True, 200), ("SIMPLS", 0.0, True, 640), ("SIMPLS", 300.0, True, 200),
    ("SIMPLS", 0.0, False, 640), ("UNIPALS", 0.0, True, 9000), ("SIMPLS", 0.0, True, 9000), ("KERNEL", 0.0, True, 9000)])
def test_several_responses_nipals_never_writes_x(gpu_required, monkeypatch, method, offset, standardize, n):
    """Several responses (NIPALS, UNIPALS) and SIMPLS without a standardised working copy: one fused raw-space sweep pass per
    component (loadings p = X^T ts and X p, everything else from maintained products); KERNEL: statistics-only
    preparation, products with the operand standardised in registers.  Earlier form of this test: ONE raw-space
    pass for the loadings, the weights of the last sweep vector and their block scores (mbpls_hip_loadings_sweep_raw:
    no deflation) -- against the oracle and the stored path, with the caller's matrix bit-identical afterwards.
    n = 9000: the long-column form of the same pass (clusters of workgroups sharing a column)."""
    import mbpls_amd
    from mbpls_amd import MBPLS, DeviceMatrix
    from mbpls_amd.engine import HipEngine
    p = 3001 if n < 8192 else (400 if method in ("NIPALS", "SIMPLS") else 9500)   # (UNIPALS / KERNEL: the p > n branches)
    X, Y = orc.make_latent(n, p, 3, seed=5)
    X = X + offset * (1.0 + np.arange(p) % 4)
    offs = [0, p // 3, p]
    Xs = [np.ascontiguousarray(X[:, a:b]) for a, b in zip(offs[:-1], offs[1:])]
    ref = orc.fit(Xs, Y, n_components=3, method=method, standardize=standardize, full_svd=True)
    used = []
    real = HipEngine.prepare_stats

    def spy(self, *a, **k):
        used.append(True)
        return real(self, *a, **k)

    monkeypatch.setattr(HipEngine, "prepare_stats", spy)
    monkeypatch.setenv("MBPLS_KERNEL_IMPLICIT", "1")
    monkeypatch.setenv("MBPLS_KERNEL_GRAM", "1")     # (KERNEL on maintained products: test_kernel_method_on_maintained_products)
    ctx = mbpls_amd.get_context()
    Xd = DeviceMatrix(ctx, n, p, offs)
    Xd.upload(X)
    mdl = MBPLS(n_components=3, method=method, standardize=standardize, full_svd=True).fit(Xd, Y)
    if method != "KERNEL" and not (method == "SIMPLS" and n > 8192):
        # one fused sweep pass per component (p = X^T t and X p), standardising in registers (tall SIMPLS keeps the products)
        assert used == [] and "never written" in mdl.fit_plan_["standardisation"], mdl.fit_plan_
    else:
        assert used == [True]
    assert np.array_equal(Xd.download(), X)
    tol = max(1e-8, 1e4 * offset * np.finfo(float).eps)
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"{method} q=3 never-writes-X") < tol
    if method == "NIPALS":
        assert [abs(a - b) <= 2 for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_)] == [True] * 3
    monkeypatch.setenv("MBPLS_KERNEL_IMPLICIT", "0")
    monkeypatch.setenv("MBPLS_IMPLICIT_STD", "0")
    used.clear()
    m2 = MBPLS(n_components=3, method=method, standardize=standardize, full_svd=True).fit(Xd, Y)
    assert used == []
    assert util.compare(util.model_to_flat(m2), util.model_to_flat(mdl), TOL, what="stored vs never-writes-X") < tol
    Xd.close()


@pytest.mark.parametrize("nodeflate", ["1", "0"])
def test_unipals_several_responses_never_deflates(gpu_required, monkeypatch, nodeflate):
    """UNIPALS, p > n, q > 1: by default X is standardised once and never deflated (X_k = P X_0, P = I - Ts Ts^T applied
    to the n-vectors around three read-only passes per component); the explicitly deflating form is a fallback
    (forced here).  Same numbers, equal to the oracle."""
    from mbpls_amd import MBPLS
    monkeypatch.setattr("mbpls_amd.plan.FitOptions.forced", {"unipals_nodeflate": nodeflate == "1"})
    X, Y = orc.make_latent(450, 3900, 3, seed=17)
    Xs = [np.ascontiguousarray(X[:, :1900]), np.ascontiguousarray(X[:, 1900:])]
    Xn, Yn = orc.make_latent(30, 3900, 3, seed=18)
    Xns = [np.ascontiguousarray(Xn[:, :1900]), np.ascontiguousarray(Xn[:, 1900:])]
    for std in (True, False):
        ref = orc.fit(Xs, Y, n_components=5, method="UNIPALS", standardize=std, full_svd=True)
        mdl = MBPLS(n_components=5, method="UNIPALS", standardize=std, full_svd=True).fit(Xs, Y)
        got = _fit_flat(mdl, Xns, Yn, block_scores=std)
        rflat = util.model_to_flat(ref)
        rflat["predict"] = orc.predict(ref, Xns)
        if std:
            rTs, rT, rU = orc.transform(ref, Xns, Yn, return_block_scores=True)
            for b, blk in enumerate(rT):
                rflat[f"T_new/{b}"] = blk
        else:
            rTs, rU = orc.transform(ref, Xns, Yn)
        rflat["Ts_new"], rflat["U_new"] = rTs, rU
        assert util.compare(got, rflat, TOL, what=f"unipals nodeflate={nodeflate} std={std}") < 1e-8


@pytest.mark.parametrize("subspace,gram", [("1", "1"), ("0", "1"), ("0", "0")])
def test_multi_response_nipals_on_association_matrices(gpu_required, monkeypatch, subspace, gram):
    """Several responses: the NIPALS inner loop runs (default) in the q-dimensional space of its iterates -- u = Y c,
    ts = G c / ||G c|| with G = P X (X^T (P Y)), the whole loop of a component in one kernel launch -- or
    (fallback forms, forced) on the per-block association matrices A_b = X_b X_b^T (t_b = A_b u / sqrt(u^T A_b u)),
    instead of sweeping X every iteration: same iterates, same results as the X-sweeping loop (both switches 0) and as
    the oracle."""
    from mbpls_amd import MBPLS
    monkeypatch.setattr("mbpls_amd.plan.FitOptions.forced", {"nipals_subspace": subspace == "1", "nipals_gram": gram == "1"})
    X, Y = orc.make_latent(333, 4000, 4, seed=90)
    Xs = [np.ascontiguousarray(X[:, :1500]), np.ascontiguousarray(X[:, 1500:2501]), np.ascontiguousarray(X[:, 2501:])]
    Xn, Yn = orc.make_latent(40, 4000, 4, seed=91)
    Xns = [np.ascontiguousarray(Xn[:, :1500]), np.ascontiguousarray(Xn[:, 1500:2501]), np.ascontiguousarray(Xn[:, 2501:])]
    for std in (True, False):
        ref = _cached_oracle(("association", std), lambda: orc.fit(Xs, Y, n_components=4, standardize=std))
        mdl = MBPLS(n_components=4, standardize=std).fit(Xs, Y)
        got = _fit_flat(mdl, Xns, Yn)
        rflat = util.model_to_flat(ref)
        rflat["predict"] = orc.predict(ref, Xns)
        rTs, rT, rU = orc.transform(ref, Xns, Yn, return_block_scores=True)
        for b, blk in enumerate(rT):
            rflat[f"T_new/{b}"] = blk
        rflat["Ts_new"], rflat["U_new"] = rTs, rU
        w = util.compare(got, rflat, TOL, what=f"subspace={subspace} gram={gram} std={std}")
        assert w < 1e-8, w
        assert all(abs(a - b) <= 2 for a, b in zip(mdl.nipals_iterations_, ref.nipals_iterations_))


@pytest.mark.parametrize("n,q,kind,k,qspace", [(300, 2, 0, 0, False), (1000, 4, 1, 2, False), (4096, 10, 0, 5, False),
                                               (8192, 16, 2, 3, False), (777, 3, 3, 1, False), (300, 2, 0, 0, True),
                                               (4096, 10, 0, 5, True), (8192, 16, 0, 3, True), (20000, 5, 0, 2, True),
                                               (64, 1, 0, 0, True)])
def test_nipals_subspace_loop_kernel(ctx, n, q, kind, k, qspace):
    """The whole multi-response NIPALS loop of a component in one launch against its numpy statement
    (solvers._subspace_loop_host): iteration count, ts, u, v, the coefficients of the last sweep vector -- on n-vectors
    (k_nipals_subspace_loop, every norm) and in q x q algebra (k_nipals_subspace_qloop, 2-norm, any n)."""
    from mbpls_amd import solvers
    from mbpls_amd.engine import HipEngine
    rng = np.random.RandomState(n + q)
    Yl = rng.standard_normal((n, q)); Yl -= Yl.mean(axis=0)
    pl = min(3 * n // 2 + 7, 3000)
    Xl = rng.standard_normal((n, pl)) * (0.97 ** np.arange(pl))
    Tp = np.linalg.qr(rng.standard_normal((n, k)))[0] if k else np.zeros((n, 0))
    PY = Yl - Tp @ (Tp.T @ Yl)
    G = Xl @ (Xl.T @ PY)
    G = G - Tp @ (Tp.T @ G)
    c0 = np.zeros((q, 1)); c0[0, 0] = 1.0
    norm = {0: 2, 1: 1, 2: np.inf, 3: -np.inf}[kind]
    want = solvers._subspace_loop_host(G, Yl, c0, norm, 1e-14, 5000)
    eng = HipEngine(ctx=ctx)
    eng.n = n
    eng.opts.nipals_qspace = qspace
    got = eng.nipals_subspace_loop(G, Yl, c0, kind, 1e-14, 5000)
    eng._ss and [a.free() for key, a in eng._ss.items() if hasattr(a, "free")]
    assert got is not None
    assert abs(got[0] - want[0]) <= max(2, want[0] // 50) and got[1] == want[1], (got[0], want[0])
    for a, b, nm in zip(got[2:], want[2:], ("ts", "u", "v", "c_used")):
        assert util.relerr(a, b) < 1e-9, nm


@pytest.mark.parametrize("n,k,q,kind", [(257, 0, 1, 0), (1000, 3, 4, 0), (4096, 9, 10, 1), (5000, 16, 16, 2), (300, 2, 3, 3)])
def test_nipals_gram_step_kernel(ctx, n, k, q, kind):
    """One device iteration of the association-matrix NIPALS loop against its numpy statement, incl. the done flag."""
    from mbpls_amd import _capi
    rng = np.random.RandomState(n + k + q)
    Tp = np.linalg.qr(rng.standard_normal((n, max(k, 1))))[0][:, :k]
    Y = rng.standard_normal((n, q))
    a = rng.standard_normal(n)
    ts_old = rng.standard_normal(n); ts_old /= np.linalg.norm(ts_old)
    Pa = a - Tp.dot(Tp.T.dot(a))
    ts = Pa / np.linalg.norm(Pa)
    d = ts_old - ts
    diff = [np.linalg.norm(d), np.abs(d).sum(), np.abs(d).max(), np.abs(d).min()][kind]
    v = Y.T.dot(ts) / ts.dot(ts)
    u = Y.dot(v) / v.dot(v); u /= np.linalg.norm(u)
    z = u - Tp.dot(Tp.T.dot(u))
    B = 2
    dts, du, dv, dz, dst = ctx.from_host(ts_old), ctx.empty(n), ctx.empty(q), ctx.empty(n * B), ctx.from_host(np.array([0.0, 4.0, 0.0]))
    da, dTp, dY = ctx.from_host(a), (ctx.from_host(Tp) if k else None), ctx.from_host(np.asfortranarray(Y))

    def step(tol):
        _capi.check(ctx.lib.mbpls_hip_nipals_gram_step(ctx.handle, n, da.ptr, dTp.ptr if k else None, k, dY.ptr, q, kind,
                                                       tol, 100, dts.ptr, du.ptr, dv.ptr, dz.ptr, B, dst.ptr))
    step(diff * 2)          # converges: 5th iteration, diff below tol
    st = dst.get()
    assert st[0] == 1.0 and st[1] == 5.0 and abs(st[2] - diff) <= 1e-12 * max(1.0, diff)
    assert util.relerr(dts.get(), ts) < 1e-12 and util.relerr(du.get(), u) < 1e-12 and util.relerr(dv.get(), v) < 1e-12
    zz = dz.get()
    assert util.relerr(zz[:n], z) < 1e-10 and np.array_equal(zz[:n], zz[n:])
    da.set(rng.standard_normal(n))
    step(diff * 2)          # done: a queued step must not change anything
    assert np.array_equal(dst.get(), st) and util.relerr(dts.get(), ts) < 1e-12


def test_refit_never_touches_what_the_caller_holds(gpu_required):
    """Every fit hands out freshly allocated attribute arrays (their pages are mapped by a helper thread while the GPU
    works, engine._prefault): whatever the caller kept from an earlier fit of the same estimator stays as it was."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(200, 150_000, 1, seed=3)
    X2, Y2 = orc.make_latent(200, 150_000, 1, seed=4)
    m = MBPLS(n_components=3).fit(X, Y)
    held = [m.P_[0], m.R_, m.W_[0], m.beta_, m.x_scalers_[0].scale_]
    copies = [a.copy() for a in held]
    m.fit(X2, Y2)
    for a, c in zip(held, copies):
        assert np.array_equal(a, c)                           # still fit 1's numbers
    now = [m.P_[0], m.R_, m.W_[0], m.beta_, m.x_scalers_[0].scale_]
    assert all(not np.shares_memory(a, b) for a in held for b in now)
    ref = orc.fit(X2, Y2, n_components=3)
    assert util.compare(util.model_to_flat(m), util.model_to_flat(ref), TOL, what="refit") < 1e-9


def test_no_device_memory_growth(gpu_required):
    """Every fit / predict / transform returns its device buffers (working copies, slots, association matrices,
    loop state) to the context pool: free device memory after trimming the pool is the same after many more fits."""
    import gc
    import mbpls_amd
    from mbpls_amd import MBPLS
    ctx = mbpls_amd.get_context()
    X, Y = orc.make_latent(300, 6000, 3, seed=1)
    Xs = [np.ascontiguousarray(X[:, :2500]), np.ascontiguousarray(X[:, 2500:])]
    Xn = X.copy(); Xn[np.random.RandomState(0).rand(*X.shape) < 0.02] = np.nan

    def one_round():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for method in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
                for q in (1, 3):
                    m = MBPLS(n_components=3, method=method).fit(Xs, Y[:, :q])
                    m.predict(Xs); m.transform(Xs, Y[:, :q])
            MBPLS(n_components=2, sparse_data=True).fit(Xn, Y[:, :1]).predict(Xn)

    def free_bytes():
        gc.collect()
        ctx.trim_pool()
        return ctx.info()["free_bytes"]

    one_round()
    before = free_bytes()
    for _ in range(3):
        one_round()
    assert before - free_bytes() < (8 << 20)


@pytest.mark.parametrize("mbytes", [0.2, 1, 9, 40, 100])
def test_large_host_to_device_copies_round_trip(gpu_required, mbytes):
    """Uploads above 256 KB bounce through the context's ring of four pinned 8 MB slots (ctx.hip): single chunk, several
    chunks, and more chunks than slots (the ring wraps and waits for the DMA that last read a slot) -- also back to back
    with the source overwritten right after the call returns, which the bounce makes legal."""
    import mbpls_amd
    ctx = mbpls_amd.get_context()
    n = int(mbytes * (1 << 20) / 8)
    rng = np.random.default_rng(int(mbytes * 10))
    a = rng.standard_normal(n)
    d1, d2 = ctx.empty(n), ctx.empty(n)
    d1.set(a)
    b = a.copy()
    d2.set(b)
    b[:] = -1.0                                   # the call has returned: the source may be reused
    assert np.array_equal(d1.get(), a)
    assert np.array_equal(d2.get(), a)
    d1.free(); d2.free()


@pytest.mark.parametrize("n", [1025, 1200, 1280, 1281, 1536, 2200, 2560, 2561, 3000, 3072, 3073])
def test_ragged_sample_counts_below_4096(gpu_required, n):
    """1024 < n < 4096: the statistics pass picks sixteen, twenty or twenty-four rows per thread by the wave-instructions
    a column costs and by whether its waves divide the eight slots of a CU (colpass_util.h pick_cfg: one wave of twenty
    rows at n = 1200, two of twenty at 2500, two of twenty-four at 3000; n = 1280 / 2560 / 3072 fill those exactly, one
    more sample does not), the other passes stay on sixteen / eight -- offset data, two blocks, against the oracle."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(n, 900, 1, seed=n)
    X = X + 2.0
    blocks = [np.ascontiguousarray(X[:, :333]), np.ascontiguousarray(X[:, 333:])]
    ref = orc.fit(blocks, Y, n_components=3, method="NIPALS")
    mdl = MBPLS(n_components=3).fit(blocks, Y)
    assert "never written" in str(mdl.fit_plan_.get("standardisation")), mdl.fit_plan_
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"ragged n={n}") < 1e-9
    assert util.relerr(mdl.predict(blocks), orc.predict(ref, blocks)) < 1e-9
    mu = MBPLS(n_components=3, method="UNIPALS").fit(blocks, Y)
    ru = orc.fit(blocks, Y, n_components=3, method="UNIPALS", full_svd=True)
    assert util.compare(util.model_to_flat(mu), util.model_to_flat(ru), TOL, what=f"ragged n={n}, UNIPALS") < 1e-9


@pytest.mark.parametrize("n", [4097, 4500, 5119, 5120, 5121, 6001, 6144, 6145, 8191])
def test_ragged_sample_counts_above_4096(gpu_required, monkeypatch, n):
    """4096 < n < 8192: columns that do not fill their workgroup (loads without selects, masked where a column becomes the
    current one; up to n = 5120 ten rows per thread for the single-reduction passes; twenty / twenty-four rows on 256
    threads for the statistics pass up to n = 5120 / 6144, which fill them exactly; sixteen otherwise), odd n (one padding
    element per column), never-writes-X and stored / explicitly deflating passes -- against the oracle."""
    from mbpls_amd import MBPLS
    X, Y = orc.make_latent(n, 1500, 1, seed=n)
    X = X + 3.0
    blocks = [np.ascontiguousarray(X[:, :700]), np.ascontiguousarray(X[:, 700:])]
    ref = orc.fit(blocks, Y, n_components=3, method="NIPALS")
    mdl = MBPLS(n_components=3).fit(blocks, Y)
    util.check_nipals_iterations(mdl, ref, 3)
    assert util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), TOL, what=f"ragged n={n}") < 1e-9
    assert util.relerr(mdl.predict(blocks), orc.predict(ref, blocks)) < 1e-9
    monkeypatch.setenv("MBPLS_DEFLATION", "explicit")
    monkeypatch.setenv("MBPLS_IMPLICIT_STD", "0")
    m2 = MBPLS(n_components=3).fit(blocks, Y)
    assert util.compare(util.model_to_flat(m2), util.model_to_flat(ref), TOL, what=f"ragged n={n}, stored") < 1e-9
    mu = MBPLS(n_components=3, method="UNIPALS").fit(blocks, Y)
    ru = orc.fit(blocks, Y, n_components=3, method="UNIPALS", full_svd=True)
    assert util.compare(util.model_to_flat(mu), util.model_to_flat(ru), TOL, what=f"ragged n={n}, UNIPALS") < 1e-9


def test_simpls_folds_without_standardisation_are_not_batched(gpu_required):
    """With one response the four methods fit the same model -- on CENTRED data.  SIMPLS centres its scores itself
    (mbpls.py:1007-1009), so with standardize=False it is a different model from the others and its folds must not take
    the batched NIPALS passes (they did: found by the randomised sweep of round 5)."""
    from mbpls_amd import MBPLS, cv_batch
    from mbpls_amd.model_selection import cross_val_predict
    n, p, K = 60, 40, 3
    X, Y = orc.make_latent(n, p, 1, r=4, seed=3)
    X = X + 2.0                                        # not centred
    est = MBPLS(n_components=K, method="SIMPLS", standardize=False, full_svd=True)
    assert not cv_batch.eligible(est, 1, n, p, 5)
    assert cv_batch.eligible(MBPLS(n_components=K, method="SIMPLS", full_svd=True), 1, n, p, 5)
    assert cv_batch.eligible(MBPLS(n_components=K, method="KERNEL", standardize=False), 1, n, p, 5)
    from sklearn.model_selection import KFold
    want = np.empty_like(Y)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        got = cross_val_predict(est, X, Y, cv=5)
        for tr, te in KFold(5).split(X):
            r = orc.fit(X[tr], Y[tr], n_components=K, method="SIMPLS", standardize=False, full_svd=True)
            want[te] = orc.predict(r, X[te])
        # the other three methods DO coincide on such data, batched
        k_b = cross_val_predict(MBPLS(n_components=K, method="KERNEL", standardize=False), X, Y, cv=5)
        k_l = cross_val_predict(MBPLS(n_components=K, method="KERNEL", standardize=False), X, Y, cv=5, batch_folds=False)
    assert util.relerr(got, want) < 1e-8
    assert util.relerr(k_b, k_l) < 1e-9


@pytest.mark.parametrize("centred", [True, False])
@pytest.mark.parametrize("n", [12, 36, 128, 300])
def test_simpls_follows_the_reference_to_its_own_conditioning(gpu_required, n, centred):
    """SIMPLS with six to twelve components.  The reference orthogonalises the new loading against the earlier ones by
    CLASSICAL Gram-Schmidt (v = p - V V^T p, mbpls.py:1015-1018) and never re-orthogonalises the scores, so ITS OWN
    scores lose orthogonality: |Ts^T Ts - I| of the oracle runs from 1e-12 to 1e-2 here (uncentred data, K close to n).
    Rounding differences between two faithful implementations are amplified by the same factor; what can be asked of
    the HIP path is therefore
        relerr(GPU, oracle) <= max(1e-8, 50 x the oracle's own defect),   and <= 1e-6 wherever that defect is < 1e-9
    over every fitted attribute -- the randomised sweep (tests/fuzz.py) applies the same criterion instead of capping K."""
    from mbpls_amd import MBPLS
    seen = 0
    for K in range(6, 13):
        if K >= n:
            continue
        for q, p in ((1, 60), (3, 2500), (2, 400)):
            X, Y = orc.make_latent(n, p, q, r=min(4, p), seed=n + K + q)
            if not centred:
                X = X + np.random.RandomState(K).uniform(-3, 3, size=p)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ref = orc.fit(X, Y, n_components=K, method="SIMPLS", standardize=centred, full_svd=True)
                mdl = MBPLS(n_components=K, method="SIMPLS", standardize=centred, full_svd=True).fit(X, Y)
            defect = float(np.max(np.abs(ref.Ts_.T.dot(ref.Ts_) - np.eye(K))))
            w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), np.inf)
            what = f"n={n} p={p} q={q} K={K} centred={centred}: deviation {w:.2e}, oracle's |Ts^T Ts - I| {defect:.2e}"
            assert w <= max(1e-8, 50.0 * defect), what
            if defect < 1e-9:
                assert w <= 1e-6, what
                seen += 1
    assert seen >= (5 if centred else 1)       # (the well-conditioned cases are not the exception)
