"""Pins the CPU oracle (both modes) to the reference: its own 69 golden CSVs and
fixtures captured from the imported reference (tools/make_golden.py)."""
import numpy as np
import pytest

from oracle import mbpls_oracle as orc
from tests import util

TOL = 1e-9   # oracle vs reference-generated fixtures (observed <= ~1e-12)


def _flat_with_new(model, Xnew, Ynew, block_scores=True):
    flat = util.model_to_flat(model)
    flat["predict"] = orc.predict(model, Xnew)
    if model["method"] != "SIMPLS" and block_scores:
        Ts, T, U = orc.transform(model, Xnew, Ynew, return_block_scores=True)
        for b, blk in enumerate(T):
            flat[f"T_new/{b}"] = blk
    else:
        Ts, U = orc.transform(model, Xnew, Ynew)
    flat["Ts_new"], flat["U_new"] = Ts, U
    return flat


@pytest.mark.parametrize("faithful", [False, True])
@pytest.mark.parametrize("fixture,methods", [
    ("ref_golden_pgtn", ["UNIPALS", "NIPALS", "KERNEL", "SIMPLS"]),
    ("ref_golden_ngtp", ["UNIPALS", "KERNEL"]),
])
def test_reference_csvs(fixture, methods, faithful):
    """The reference's own golden CSVs (mbpls/tests/test_mbpls.py:66-117):
    absolute-value comparison with np.allclose defaults, exactly as it does."""
    g = util.load(fixture)
    X = [g["x1_train"], g["x2_train"]]
    Xt = [g["x1_test"], g["x2_test"]]
    preds = []
    for m in methods:
        for full in (True, False):
            mdl = orc.fit(X, g["y_train"], n_components=2, method=m, standardize=True,
                          full_svd=full, faithful=faithful)
            csv = util.sub(g, f"csv/{m}")
            got = {"Ts": mdl.Ts_, "U": mdl.U_, "V": mdl.V_, "beta": mdl.beta_,
                   "P1": mdl.P_[0], "P2": mdl.P_[1], "Y_predict_test": orc.predict(mdl, Xt)}
            if m != "SIMPLS":
                Ts_t, T_t, U_t = orc.transform(mdl, Xt, g["y_test"], return_block_scores=True)
                got.update(T=np.hstack(mdl.T_), A=mdl.A_, T_test=np.hstack(T_t))
            else:
                Ts_t, U_t = orc.transform(mdl, Xt, g["y_test"])
            got.update(Ts_test=Ts_t, U_test=U_t)
            assert set(got) == set(csv)
            for k, ref in csv.items():
                assert np.allclose(np.abs(got[k]), np.abs(ref)), (m, k)
                assert util.relerr(np.abs(got[k]), np.abs(ref)) < 1e-9, (m, k)
            # and everything the CSVs omit, against the captured reference attributes
            ref = util.sub(g, f"ref/{m}/full{int(full)}")
            util.compare(_flat_with_new(mdl, Xt, g["y_test"]), ref, TOL, what=f"{fixture}/{m}")
        preds.append(got["Y_predict_test"])
    for p in preds:  # mbpls/tests/test_mbpls.py:121-123
        assert np.allclose(preds[0], p, atol=1e-3)


@pytest.mark.parametrize("faithful", [False, True])
def test_quickstart_config1(faithful):
    g = util.load("quickstart_c1")
    for m in ("NIPALS", "UNIPALS", "KERNEL", "SIMPLS"):
        mdl = orc.fit(g["x"], g["y"], n_components=2, method=m, full_svd=(m == "SIMPLS"), faithful=faithful)
        util.compare(_flat_with_new(mdl, g["x"], g["y"]), util.sub(g, f"ref/{m}"), TOL, what=m)
        if m == "NIPALS":
            assert mdl.nipals_iterations_ == [2, 2]


@pytest.mark.parametrize("faithful", [False, True])
@pytest.mark.parametrize("fixture", ["matrix_pgtn", "matrix_ngtp"])
def test_matrix(fixture, faithful):
    g = util.load(fixture)
    splits = np.cumsum(g["splits"])[:-1]
    n_cases = 0
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
                        assert ref, pre
                        mdl = orc.fit(sp(X), Y, n_components=3, method=m, standardize=std,
                                      calc_all=ca, full_svd=(m == "SIMPLS"), faithful=faithful)
                        has_T_new = any(k.startswith("T_new") for k in ref)
                        got = _flat_with_new(mdl, sp(Xn), Yn, block_scores=has_T_new)
                        util.compare(got, ref, TOL, what=pre)
                        n_cases += 1
    assert n_cases == 64


def test_edges():
    g = util.load("edges")
    # constant column: scale 1, zero weight
    mdl = orc.fit(g["const/X"], g["const/Y"], n_components=3)
    assert mdl.x_scalers_[0].scale_[5] == 1.0
    assert mdl.beta_[5, 0] == 0.0
    util.compare(_flat_with_new(mdl, g["const/X"], g["const/Y"]), util.sub(g, "ref/const"), TOL, what="const")
    X, Y = g["plain/X"], g["plain/Y"]
    m0 = orc.fit(X, Y, n_components=3)
    m1 = orc.fit(X, Y.ravel(), n_components=3)
    m2 = orc.fit([X], Y, n_components=3)
    assert np.array_equal(m0.beta_, m1.beta_) and np.array_equal(m0.beta_, m2.beta_)
    for nm, o in (("1", 1), ("inf", np.inf), ("-inf", -np.inf), ("fro", "fro"), ("nuc", "nuc"), ("-1", -1), ("-2", -2)):
        mm = orc.fit(X, Y, n_components=3, nipals_convergence_norm=o)
        assert util.relerr(mm.beta_, g[f"ref/norm/{nm}/beta_"]) < TOL
    for bad in (0, 3):
        with pytest.raises(ValueError):
            orc.fit(X, Y, n_components=2, nipals_convergence_norm=bad)
    with pytest.raises(NameError):
        orc.fit(X, Y, method="nope")
    X3, Y3 = g["q3/X"], g["q3/Y"]
    for nm, o in (("2", 2), ("1", 1), ("inf", np.inf)):
        mm = orc.fit([X3[:, :50], X3[:, 50:]], Y3, n_components=3, nipals_convergence_norm=o)
        util.compare(_flat_with_new(mm, [X3[:, :50], X3[:, 50:]], Y3), util.sub(g, f"ref/q3/norm{nm}"),
                     1e-8, what=f"q3/{nm}")


def test_scaler_matches_fixture():
    g = util.load("matrix_pgtn")
    X = g["in/q1/X"]
    s = orc.ColumnScaler().fit(X)
    pre = "ref/q1/single/NIPALS/std1/ca1"
    assert np.array_equal(s.mean_, g[pre + "/x_mean/0"])
    assert util.relerr(s.scale_, g[pre + "/x_scale/0"]) < 1e-15


def test_sparse_oracle_vs_reference_fixture():
    """sparse_data=True: the NaN-aware oracle against attributes captured from the imported reference."""
    from oracle import mbpls_oracle_sparse as ors
    g = util.load("sparse")
    n_cases = 0
    for q in (1, 2):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        Xn, Yn = g[f"in/q{q}/Xnew"], g[f"in/q{q}/Ynew"]
        for blocks in ("single", "multi"):
            sp = (lambda A: A) if blocks == "single" else (lambda A: [A[:, :40], A[:, 40:]])
            for std in (True, False):
                ref = util.sub(g, f"ref/q{q}/{blocks}/std{int(std)}")
                mdl = ors.fit(sp(X), Y, n_components=3, standardize=std)
                flat = util.model_to_flat(mdl)
                flat["predict"] = ors.predict(mdl, sp(Xn))
                Ts, T, U = ors.transform(mdl, sp(Xn), Yn, return_block_scores=True)
                flat["Ts_new"], flat["U_new"] = Ts, U
                for b, blk in enumerate(T):
                    flat[f"T_new/{b}"] = blk
                util.compare(flat, ref, 1e-8, what=f"sparse q{q} {blocks} std{std}")
                n_cases += 1
    assert n_cases == 8


def test_sparse_oracle_with_missing_responses_vs_reference_fixture():
    """sparse_data=True with NaN in Y as well (mbpls.py:832-836, :890-909; transform :1156-1166): the oracle against
    attributes captured from the imported reference on patterns the reference handles (missing responses only in
    sparse rows of the last X block)."""
    from oracle import mbpls_oracle_sparse as ors
    g = util.load("sparse_y")
    n_cases = 0
    for q in (2, 3):
        X, Y = g[f"in/q{q}/X"], g[f"in/q{q}/Y"]
        Xn, Yn = g[f"in/q{q}/Xnew"], g[f"in/q{q}/Ynew"]
        assert np.isnan(Y[:, 0]).any() and np.isnan(Yn).any()
        for blocks in ("single", "multi"):
            sp = (lambda A: A) if blocks == "single" else (lambda A: [A[:, :40], A[:, 40:]])
            for std in (True, False):
                ref = util.sub(g, f"ref/q{q}/{blocks}/std{int(std)}")
                mdl = ors.fit(sp(X), Y, n_components=3, standardize=std)
                flat = util.model_to_flat(mdl)
                flat["predict"] = ors.predict(mdl, sp(Xn))
                Ts, T, U = ors.transform(mdl, sp(Xn), Yn, return_block_scores=True)
                flat["Ts_new"], flat["U_new"] = Ts, U
                for b, blk in enumerate(T):
                    flat[f"T_new/{b}"] = blk
                util.compare(flat, ref, 1e-8, what=f"sparse_y q{q} {blocks} std{std}")
                if std:
                    assert util.relerr(mdl.y_scaler_.mean_, ref["y_mean"]) < 1e-13
                    assert util.relerr(mdl.y_scaler_.scale_, ref["y_scale"]) < 1e-13
                n_cases += 1
    assert n_cases == 8
    # a missing response in a row the reference does not patch: nothing to restate
    Yb = g["in/q2/Y"].copy()
    dense_rows = np.where(~np.isnan(g["in/q2/X"]).any(axis=1))[0]
    Yb[dense_rows[0], 1] = np.nan
    with pytest.raises(ValueError):
        ors.fit(g["in/q2/X"], Yb, n_components=2)


def _loo(fit, predict, X, y, K):
    """leave-one-out predictions by a loop of fits (what sklearn's cross_val_predict(est, X, y, cv=len(y)) does)"""
    n = len(y)
    out = np.empty(n)
    for i in range(n):
        keep = np.arange(n) != i
        Xtr = [b[keep] for b in X] if isinstance(X, list) else X[keep]
        Xte = [b[i:i + 1] for b in X] if isinstance(X, list) else X[i:i + 1]
        out[i] = np.ravel(predict(fit(Xtr, y[keep].reshape(-1, 1), K), Xte))[0]
    return out


@pytest.mark.parametrize("faithful", [False, True])
def test_reference_real_data(faithful):
    """The reference's bundled REAL data through the work flows of its example notebooks (fixture real_pectin,
    tools/make_golden.py: FTIR spectra 37 x 148 + carbohydrate microarrays 37 x 30 -> pectin yield): the two-block fit,
    single-block PLS of either block, and the leave-one-out curve over 1 .. 15 components of 36 samples each -- real
    spectra are collinear, the deep components are the least well conditioned fits in the suite."""
    g = util.load("real_pectin")
    X1, X2, y = g["x1"], g["x2"], g["y"]
    Y = y.reshape(-1, 1)
    m = orc.fit([X1, X2], Y, n_components=3, faithful=faithful)
    util.compare(_flat_with_new(m, [X1, X2], Y), util.sub(g, "ref/two_blocks"), TOL, what="real data, two blocks")
    for name, X in (("ftir", X1), ("carb", X2)):
        m = orc.fit(X, Y, n_components=5, faithful=faithful)
        util.compare(_flat_with_new(m, X, Y), util.sub(g, f"ref/{name}"), 1e-8, what=f"real data, {name}")
    fit = lambda X, Y, K: orc.fit(X, Y, n_components=K, faithful=faithful)
    X12 = np.hstack((X1, X2))
    for k in (1, 3, 6, 10, 15):
        got = _loo(fit, orc.predict, X12, y, k)
        assert util.relerr(got, g["loo/x1x2"][:, k - 1]) < 1e-7, (k, util.relerr(got, g["loo/x1x2"][:, k - 1]))
    for name, X in (("ftir", X1), ("carb", X2)):
        for c, k in enumerate((1, 3, 5)):
            got = _loo(fit, orc.predict, X, y, k)
            assert util.relerr(got, g[f"loo/{name}"][:, c]) < 1e-7, (name, k)


@pytest.mark.parametrize("faithful", [False, True])
def test_reference_mbdata(faithful):
    """mbpls/data/MBdata.mat (fixture mbdata, tools/make_golden.py): ten samples x 100 variables in two or four blocks,
    three responses -- every method on both layouts, standardised and not -- and the same matrix with four missing
    values through sparse_data=True."""
    from oracle import mbpls_oracle_sparse as ors
    g = util.load("mbdata")
    X, Y, Xmv = g["X"], g["Y"], g["Xmv"]
    cases = 0
    for lay in ("Xin2", "Xin4"):
        blocks = lambda A: [np.ascontiguousarray(A[:, lo - 1:hi]) for lo, hi in g[lay]]
        for method, full in (("NIPALS", False), ("UNIPALS", True), ("KERNEL", True), ("SIMPLS", True)):
            for std in (True, False):
                m = orc.fit(blocks(X), Y, n_components=3, method=method, standardize=std, full_svd=full,
                            faithful=faithful)
                ref = util.sub(g, f"ref/{lay}/{method}/std{int(std)}")
                has_blocks = any(k.startswith("T_new") for k in ref)
                util.compare(_flat_with_new(m, blocks(X), Y, block_scores=has_blocks), ref, 1e-8,
                             what=f"mbdata {lay} {method} std{std}")
                cases += 1
        for std in (True, False):
            m = ors.fit(blocks(Xmv), Y, n_components=2, standardize=std)
            flat = util.model_to_flat(m)
            flat["predict"] = ors.predict(m, blocks(Xmv))
            Ts, T, U = ors.transform(m, blocks(Xmv), Y, return_block_scores=True)
            flat["Ts_new"], flat["U_new"] = Ts, U
            for b, blk in enumerate(T):
                flat[f"T_new/{b}"] = blk
            util.compare(flat, util.sub(g, f"sparse/{lay}/std{int(std)}"), 1e-8, what=f"mbdata sparse {lay} std{std}")
            cases += 1
    assert cases == 20
