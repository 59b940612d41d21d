#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference.

Runs only in the build container (``/root/reference`` is mounted there, read
only).  Nothing of the reference's source is copied: the fixtures are data --
captured input arrays, the reference's fitted attributes / predictions for them,
and the numeric contents of the reference's own golden CSVs
(``mbpls/tests/test_data/*.csv``, enumerated in ``mbpls/tests/load_data.py:17-86``).

    python tools/make_golden.py            # writes tests/golden/*.npz and prints pin report

Fixture sets follow SURVEY.md 8(c'):
  G0 ref_golden_pgtn   the reference's p>n golden test (tests/test_mbpls.py:34-215)
  G1 ref_golden_ngtp   the reference's n>p golden test (:220-421)
  G2 quickstart_c1     README quick-start = BASELINE config 1
  G3 matrix_pgtn       method x blocks x q x standardize x calc_all, p>n
  G4 matrix_ngtp       n>=p branches
  G5 edges             constant column, 1-D Y, list-of-one, norm variants
  G6 sparse            sparse_data=True (NaN-aware NIPALS), dense Y
  G7 sparse_y          sparse_data=True with missing responses as well
  G8 real_pectin       the reference's bundled real data (FTIR spectra 37 x 148, carbohydrate microarrays 37 x 30,
                       pectin yield) through the work flows of its example notebooks: the two-block fit, the
                       leave-one-out curves over 1 .. 15 components, single-block PLS of either block
  G9 mbdata            mbpls/data/MBdata.mat (10 samples x 100 variables in 2 or 4 blocks, three responses, and the same
                       matrix with four missing values): every method on both block layouts, sparse_data=True
"""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

warnings.filterwarnings("ignore")
from mbpls.mbpls import MBPLS  # noqa: E402  (the reference, imported in place)
from oracle import mbpls_oracle as orc  # noqa: E402

ATTRS = ["Ts_", "U_", "V_", "R_", "beta_", "A_", "A_corrected_", "explained_var_x_",
         "explained_var_y_", "explained_var_xblocks_", "W_concat_"]
LIST_ATTRS = ["T_", "W_", "P_", "W_non_normal_"]


def capture(m, store, prefix, Xnew=None, Ynew=None):
    """Copy every fitted attribute of a reference estimator into ``store``."""
    for a in ATTRS:
        if hasattr(m, a):
            store[f"{prefix}/{a}"] = np.asarray(getattr(m, a), dtype=np.float64)
    for a in LIST_ATTRS:
        if hasattr(m, a):
            v = getattr(m, a)
            if isinstance(v, list):
                for b, blk in enumerate(v):
                    store[f"{prefix}/{a}/{b}"] = np.asarray(blk, dtype=np.float64)
            else:  # SIMPLS: W_ is one array
                store[f"{prefix}/{a}"] = np.asarray(v, dtype=np.float64)
    if hasattr(m, "x_scalers_"):
        for b, s in enumerate(m.x_scalers_):
            store[f"{prefix}/x_mean/{b}"] = s.mean_.copy()
            store[f"{prefix}/x_scale/{b}"] = s.scale_.copy()
        store[f"{prefix}/y_mean"] = m.y_scaler_.mean_.copy()
        store[f"{prefix}/y_scale"] = m.y_scaler_.scale_.copy()
    if Xnew is not None:
        store[f"{prefix}/predict"] = m.predict(Xnew)
        if m.method != "SIMPLS":
            try:
                Ts, T, U = m.transform(Xnew, Ynew, return_block_scores=True)
                for b, blk in enumerate(T):
                    store[f"{prefix}/T_new/{b}"] = blk
            except UnboundLocalError:
                # reference bug (mbpls.py:1256-1286): non-standardised transform with
                # 2-D Y never builds block scores; capture without them
                Ts, U = m.transform(Xnew, Ynew)
        else:
            Ts, U = m.transform(Xnew, Ynew)
        store[f"{prefix}/Ts_new"] = Ts
        store[f"{prefix}/U_new"] = U


def ref_test_inputs(num_samples, seed=25, n1=25, n2=45, noise=5):
    """The input recipe of mbpls/tests/test_mbpls.py:36-57 with the legacy
    ortho_group sampler (the one the CSVs were produced with)."""
    np.random.seed(seed)
    l1 = np.expand_dims(np.random.randint(0, 10, n1), 1)
    l2 = np.expand_dims(np.sin(np.linspace(0, 5, n2)), 1)
    y = orc.legacy_ortho_group(num_samples, seed)[:, :2]
    x1 = np.dot(y[:, 0:1], l1.T)
    x2 = np.dot(y[:, 1:2], l2.T)
    x1 = np.random.normal(x1, 0.05 * noise)
    x2 = np.random.normal(x2, 0.05 * noise)
    idx = np.random.choice(np.arange(num_samples), num_samples, replace=False)
    k = round(num_samples * 8 / 10)
    tr, te = idx[:k], idx[k:]
    return dict(x1_train=x1[tr], x2_train=x2[tr], y_train=y[tr],
                x1_test=x1[te], x2_test=x2[te], y_test=y[te])


def load_csvs(tag, methods):
    """Numeric contents of the reference's golden CSVs for one test family."""
    d = os.path.join(REF, "mbpls", "tests", "test_data")
    out = {}
    names = ["T", "A", "T_test", "Ts_test", "U_test", "P1", "P2", "Ts", "U", "V", "beta", "Y_predict_test"]
    for m in methods:
        for nm in names:
            f = os.path.join(d, f"{nm}_{tag}{m}.csv")
            if os.path.exists(f):
                out[f"csv/{m}/{nm}"] = np.genfromtxt(f, delimiter=",")
    return out


def golden_ref_tests():
    report = []
    for fname, ns, tag, methods in (("ref_golden_pgtn", 50, "", ["UNIPALS", "NIPALS", "KERNEL", "SIMPLS"]),
                                    ("ref_golden_ngtp", 150, "NlargerP_", ["UNIPALS", "KERNEL"])):
        inp = ref_test_inputs(ns)
        store = dict(inp)
        csv = load_csvs(tag, methods)
        store.update(csv)
        worst = 0.0
        for full in (True, False):
            for m in methods:
                mdl = MBPLS(n_components=2, method=m, standardize=True, full_svd=full)
                mdl.fit([inp["x1_train"], inp["x2_train"]], inp["y_train"])
                pre = f"ref/{m}/full{int(full)}"
                capture(mdl, store, pre, [inp["x1_test"], inp["x2_test"]], inp["y_test"])
                # pin: the current stack reproduces the CSVs with the legacy sampler
                chk = {"Ts": store[pre + "/Ts_"], "U": store[pre + "/U_"], "V": store[pre + "/V_"],
                       "beta": store[pre + "/beta_"], "P1": store[pre + "/P_/0"], "P2": store[pre + "/P_/1"],
                       "Y_predict_test": store[pre + "/predict"], "Ts_test": store[pre + "/Ts_new"],
                       "U_test": store[pre + "/U_new"]}
                if m != "SIMPLS":
                    chk["T"] = np.hstack([store[pre + "/T_/0"], store[pre + "/T_/1"]])
                    chk["A"] = store[pre + "/A_"]
                    chk["T_test"] = np.hstack([store[pre + "/T_new/0"], store[pre + "/T_new/1"]])
                for nm, val in chk.items():
                    ref = csv[f"csv/{m}/{nm}"]
                    worst = max(worst, float(np.max(np.abs(np.abs(val) - np.abs(ref)))))
        report.append((fname, len(csv), worst))
        np.savez_compressed(os.path.join(OUT, fname + ".npz"), **store)
    return report


def golden_quickstart():
    np.random.seed(0)
    x = np.random.rand(40, 200)
    y = np.random.rand(40, 1)
    store = dict(x=x, y=y)
    for m, full in (("NIPALS", False), ("UNIPALS", False), ("KERNEL", False), ("SIMPLS", True)):
        mdl = MBPLS(n_components=2, method=m, full_svd=full)
        mdl.fit(x, y)
        capture(mdl, store, f"ref/{m}", x, y)
    np.savez_compressed(os.path.join(OUT, "quickstart_c1.npz"), **store)


def golden_matrix(fname, n, p, splits, n_new, seeds, methods, K=3):
    store = {}
    for q in (1, 3):
        X, Y = orc.make_latent(n, p, q, r=6, seed=seeds[0])
        Xn, Yn = orc.make_latent(n_new, p, q, r=6, seed=seeds[1])
        store[f"in/q{q}/X"] = X; store[f"in/q{q}/Y"] = Y
        store[f"in/q{q}/Xnew"] = Xn; store[f"in/q{q}/Ynew"] = Yn
        for blocks in ("single", "multi"):
            def split(A):
                if blocks == "single":
                    return A
                return [np.ascontiguousarray(b) for b in np.split(A, np.cumsum(splits)[:-1], axis=1)]
            for m in methods:
                for std in (True, False):
                    for ca in (True, False):
                        full = (m == "SIMPLS")  # svds fails for q=1 SIMPLS in current scipy (mbpls.py:1003)
                        mdl = MBPLS(n_components=K, method=m, standardize=std, calc_all=ca, full_svd=full)
                        mdl.fit(split(X), Y)
                        pre = f"ref/q{q}/{blocks}/{m}/std{int(std)}/ca{int(ca)}"
                        capture(mdl, store, pre, split(Xn), Yn)
    store["splits"] = np.asarray(splits)
    np.savez_compressed(os.path.join(OUT, fname + ".npz"), **store)


def golden_edges():
    X, Y = orc.make_latent(48, 180, 1, r=6, seed=7)
    store = {}
    Xc = X.copy()
    Xc[:, 5] = 3.25
    store["const/X"] = Xc; store["const/Y"] = Y
    mdl = MBPLS(n_components=3).fit(Xc, Y)
    capture(mdl, store, "ref/const", Xc, Y)
    # 1-D Y and list-of-one give the bare-array result
    m1 = MBPLS(n_components=3).fit(X, Y.ravel())
    m2 = MBPLS(n_components=3).fit([X], Y)
    m0 = MBPLS(n_components=3).fit(X, Y)
    store["plain/X"] = X; store["plain/Y"] = Y
    capture(m0, store, "ref/plain", X, Y)
    assert np.array_equal(m0.beta_, m1.beta_) and np.array_equal(m0.beta_, m2.beta_)
    for nm, o in (("1", 1), ("inf", np.inf), ("-inf", -np.inf), ("fro", "fro"), ("nuc", "nuc"), ("-1", -1), ("-2", -2)):
        mm = MBPLS(n_components=3, nipals_convergence_norm=o).fit(X, Y)
        store[f"ref/norm/{nm}/beta_"] = mm.beta_
        store[f"ref/norm/{nm}/Ts_"] = mm.Ts_
    # multiblock q=3 NIPALS iteration-heavy case with non-default norms
    X3, Y3 = orc.make_latent(40, 120, 3, r=6, seed=11)
    store["q3/X"] = X3; store["q3/Y"] = Y3
    for nm, o in (("2", 2), ("1", 1), ("inf", np.inf)):
        mm = MBPLS(n_components=3, nipals_convergence_norm=o).fit([X3[:, :50], X3[:, 50:]], Y3)
        capture(mm, store, f"ref/q3/norm{nm}", [X3[:, :50], X3[:, 50:]], Y3)
    np.savez_compressed(os.path.join(OUT, "edges.npz"), **store)


def golden_sparse():
    """sparse_data=True (NaN-aware NIPALS): no case in the reference's own tests -> pinned from the import."""
    store = {}
    for q in (1, 2):
        X, Y = orc.make_latent(40, 90, q, r=5, seed=3)
        Xn2, Yn = orc.make_latent(12, 90, q, r=5, seed=4)
        rng = np.random.RandomState(1)
        X = X.copy(); X[rng.rand(*X.shape) < 0.05] = np.nan
        Xn2 = Xn2.copy(); Xn2[rng.rand(*Xn2.shape) < 0.05] = np.nan
        store[f"in/q{q}/X"] = X; store[f"in/q{q}/Y"] = Y; store[f"in/q{q}/Xnew"] = Xn2; store[f"in/q{q}/Ynew"] = Yn
        for blocks in ("single", "multi"):
            sp = (lambda A: A.copy()) if blocks == "single" else (lambda A: [A[:, :40].copy(), A[:, 40:].copy()])
            for std in (True, False):
                mdl = MBPLS(n_components=3, sparse_data=True, standardize=std)
                mdl.fit(sp(X), Y.copy())
                pre = f"ref/q{q}/{blocks}/std{int(std)}"
                capture(mdl, store, pre)
                store[f"{pre}/predict"] = mdl.predict(sp(Xn2))
                try:
                    Ts, T, U = mdl.transform(sp(Xn2), Yn.copy(), return_block_scores=True)
                except UnboundLocalError:   # same reference bug as in capture(): std=False with 2-D Y
                    Ts, U = mdl.transform(sp(Xn2), Yn.copy())
                    Ts, T = mdl.transform(sp(Xn2), return_block_scores=True)
                store[f"{pre}/Ts_new"] = Ts; store[f"{pre}/U_new"] = U
                for b, blk in enumerate(T):
                    store[f"{pre}/T_new/{b}"] = blk
    np.savez_compressed(os.path.join(OUT, "sparse.npz"), **store)


def golden_sparse_y():
    """sparse_data=True with missing values in Y as well (mbpls.py:832-836, :890-909): the reference re-regresses the
    Y scores of the sparse rows of the LAST X block only (:903), so the missing responses sit in such rows -- anywhere
    else the reference dies with LinAlgError.  Cases: first response column sparse (start vector = first dense
    column), several sparse columns, new data with missing responses for transform."""
    store = {}
    # (the reference's masked alternating regressions do not converge for every pattern of missing responses -- its
    # loop at :841 then never ends; the seeds below are patterns on which it does)
    for q, ycols, seed in ((2, (0,), 13), (3, (0, 2), 21)):
        X, Y = orc.make_latent(60, 90, q, r=5, seed=seed)
        Xn2, Yn = orc.make_latent(14, 90, q, r=5, seed=seed + 1)
        rng = np.random.RandomState(2)
        X = X.copy(); X[rng.rand(*X.shape) < 0.015] = np.nan
        Xn2 = Xn2.copy(); Xn2[rng.rand(*Xn2.shape) < 0.03] = np.nan
        rows = np.where(np.isnan(X[:, 40:]).any(axis=1))[0]
        assert 8 <= rows.size < X.shape[0] - 5, rows.size
        Y = Y.copy(); Yn = Yn.copy()
        for i, c in enumerate(ycols):
            Y[rows[2 * i:2 * i + 6], c] = np.nan
        Yn[rng.rand(*Yn.shape) < 0.15] = np.nan
        Yn[:, -1] = np.where(np.isnan(Yn).all(axis=1), 0.3, Yn[:, -1])      # no row without any response
        store[f"in/q{q}/X"] = X; store[f"in/q{q}/Y"] = Y; store[f"in/q{q}/Xnew"] = Xn2; store[f"in/q{q}/Ynew"] = Yn
        for blocks in ("single", "multi"):
            sp = (lambda A: A.copy()) if blocks == "single" else (lambda A: [A[:, :40].copy(), A[:, 40:].copy()])
            for std in (True, False):
                mdl = MBPLS(n_components=3, sparse_data=True, standardize=std)
                mdl.fit(sp(X), Y.copy())
                assert not np.isnan(mdl.Ts_).any()
                pre = f"ref/q{q}/{blocks}/std{int(std)}"
                capture(mdl, store, pre)
                store[f"{pre}/predict"] = mdl.predict(sp(Xn2))
                try:
                    Ts, T, U = mdl.transform(sp(Xn2), Yn.copy(), return_block_scores=True)
                except UnboundLocalError:   # same reference bug as in capture(): std=False with 2-D Y
                    Ts, U = mdl.transform(sp(Xn2), Yn.copy())
                    Ts, T = mdl.transform(sp(Xn2), return_block_scores=True)
                store[f"{pre}/Ts_new"] = Ts; store[f"{pre}/U_new"] = U
                for b, blk in enumerate(T):
                    store[f"{pre}/T_new/{b}"] = blk
    np.savez_compressed(os.path.join(OUT, "sparse_y.npz"), **store)


def golden_real_pectin():
    """examples/real_world_applications/{PectinYield_MBPLS, FTIR_PLS, Carbohydrate_Microarray_PLS}.ipynb on
    mbpls/data/MBPLS_intro/*.pkl (mbpls/data/get_data.py:31-49): x1 = FTIR, x2 = carbohydrate microarrays, y = the yield
    (the index of the frames).  Captured: the data, MBPLS(3).fit([x1, x2], y) with every attribute and its scalers,
    cross_val_predict(MBPLS(k), [x1 | x2], y, cv=37) for k = 1 .. 15 (the notebooks' "find the right number of LVs"),
    the single-block fits MBPLS(5) of either block and their leave-one-out predictions."""
    import pandas as pd
    from sklearn.model_selection import cross_val_predict
    d = os.path.join(REF, "mbpls", "data", "MBPLS_intro")
    fr = {k: pd.read_pickle(os.path.join(d, k + ".pkl")) for k in
          ("ftir1", "ftir2", "ftir3", "extraction1", "extraction2", "extraction3")}
    x1 = pd.concat((fr["ftir1"], fr["ftir2"], fr["ftir3"]))
    x2 = pd.concat((fr["extraction1"], fr["extraction2"], fr["extraction3"]))
    y = np.array(x1.index, dtype=np.float64)
    X1, X2 = np.asarray(x1, dtype=np.float64), np.asarray(x2, dtype=np.float64)
    store = dict(x1=X1, x2=X2, y=y, wavenumbers=np.asarray(x1.columns, dtype=np.float64))
    m = MBPLS(n_components=3)
    m.fit([X1, X2], y)
    capture(m, store, "ref/two_blocks", [X1, X2], y.reshape(-1, 1))
    X12 = np.hstack((X1, X2))
    store["loo/x1x2"] = np.column_stack([np.ravel(cross_val_predict(MBPLS(n_components=k), X12, y, cv=len(y)))
                                         for k in range(1, 16)])
    for name, X in (("ftir", X1), ("carb", X2)):
        m = MBPLS(n_components=5)
        m.fit(X, y)
        capture(m, store, f"ref/{name}", X, y.reshape(-1, 1))
        store[f"loo/{name}"] = np.column_stack([np.ravel(cross_val_predict(MBPLS(n_components=k), X, y, cv=len(y)))
                                                for k in (1, 3, 5)])
    # everyday sklearn usage around the estimator: score / r2_score / explained_variance_score, integer input, a
    # Pipeline behind a StandardScaler, a GridSearchCV over n_components (5 folds), transform without Y
    from sklearn.model_selection import GridSearchCV
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler
    m = MBPLS(n_components=3)
    m.fit([X1, X2], y)
    store["usage/score"] = np.float64(m.score([X1, X2], y))
    store["usage/r2_score"] = np.float64(m.r2_score([X1, X2], y))
    store["usage/explained_variance_score"] = np.float64(m.explained_variance_score([X1, X2], y))
    store["usage/transform_without_y"] = np.asarray(m.transform([X1, X2]))
    store["usage/int_beta"] = MBPLS(n_components=2).fit((X1 * 1000).astype(int), y).beta_
    pipe = Pipeline([("sc", StandardScaler()), ("pls", MBPLS(n_components=2))])
    store["usage/pipeline_predict"] = np.ravel(pipe.fit(X1, y).predict(X1))
    gs = GridSearchCV(MBPLS(), {"n_components": [1, 2, 3, 4]}, cv=5, scoring="neg_mean_squared_error").fit(X1, y)
    store["usage/grid_mean_test_score"] = np.asarray(gs.cv_results_["mean_test_score"])
    store["usage/grid_best_n_components"] = np.int64(gs.best_params_["n_components"])
    # the frames go into the estimator as they are in the notebooks: the reference must give the same numbers for them
    md = MBPLS(n_components=3)
    md.fit([x1, x2], pd.DataFrame(y))
    assert np.allclose(md.beta_, store["ref/two_blocks/beta_"])
    np.savez_compressed(os.path.join(OUT, "real_pectin.npz"), **store)


def golden_mbdata():
    """mbpls/data/MBdata.mat -- a ten-sample multiblock set with its block layouts (Xin2: 2 x 50 variables, Xin4: 4 x 25),
    three responses, and Xmv, the same matrix with four entries missing.  No code of the reference reads the file; fitted
    here with the imported reference: four methods x two layouts x standardize, and sparse_data=True on Xmv."""
    from scipy.io import loadmat
    mat = loadmat(os.path.join(REF, "mbpls", "data", "MBdata.mat"))
    X, Y, Xmv = (np.asarray(mat[k], dtype=np.float64) for k in ("X", "Y", "Xmv"))
    store = dict(X=X, Y=Y, Xmv=Xmv, Xin2=np.asarray(mat["Xin2"], dtype=np.int64), Xin4=np.asarray(mat["Xin4"], dtype=np.int64))
    for lay in ("Xin2", "Xin4"):
        blocks = lambda A: [np.ascontiguousarray(A[:, lo - 1:hi]) for lo, hi in store[lay]]
        for method, full in (("NIPALS", False), ("UNIPALS", True), ("KERNEL", True), ("SIMPLS", True)):
            for std in (True, False):
                m = MBPLS(n_components=3, method=method, standardize=std, full_svd=full)
                m.fit(blocks(X), Y)
                capture(m, store, f"ref/{lay}/{method}/std{int(std)}", blocks(X), Y)
        for std in (True, False):
            m = MBPLS(n_components=2, sparse_data=True, standardize=std)
            m.fit(blocks(Xmv), Y)
            capture(m, store, f"sparse/{lay}/std{int(std)}", blocks(Xmv), Y)
    np.savez_compressed(os.path.join(OUT, "mbdata.npz"), **store)


def main():
    os.makedirs(OUT, exist_ok=True)
    rep = golden_ref_tests()
    for fname, ncsv, worst in rep:
        print(f"{fname}: {ncsv} reference CSV arrays captured; reference (this stack, legacy sampler) "
              f"vs CSV worst |abs| diff = {worst:.3e}")
        assert worst < 1e-11
    golden_quickstart()
    golden_matrix("matrix_pgtn", 48, 180, [54, 54, 72], 12, (7, 8), ["NIPALS", "UNIPALS", "KERNEL", "SIMPLS"])
    golden_matrix("matrix_ngtp", 96, 60, [18, 18, 24], 12, (9, 10), ["NIPALS", "UNIPALS", "KERNEL", "SIMPLS"])
    golden_edges()
    golden_sparse()
    golden_sparse_y()
    golden_real_pectin()
    golden_mbdata()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
