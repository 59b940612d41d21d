"""The reference's bundled real data through the work flows of its example notebooks, on the HIP path
(examples/real_world_applications/PectinYield_MBPLS.ipynb, FTIR_PLS.ipynb, Carbohydrate_Microarray_PLS.ipynb; fixture
tests/golden/real_pectin.npz captured from the imported reference by tools/make_golden.py): FTIR spectra (37 x 148) and
carbohydrate microarrays (37 x 30) against pectin yield -- 37 samples, collinear spectra, up to 15 components of 36
samples in the leave-one-out curve: the everyday shape of the package and the least well conditioned fits of the suite."""
import numpy as np
import pytest

from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def real():
    return util.load("real_pectin")


def _flat_with_new(m, Xnew, Ynew):
    flat = util.model_to_flat(m)
    flat["predict"] = m.predict(Xnew)
    Ts, T, U = m.transform(Xnew, Ynew, return_block_scores=True)
    for b, blk in enumerate(T):
        flat[f"T_new/{b}"] = blk
    flat["Ts_new"], flat["U_new"] = Ts, U
    return flat


def test_two_block_fit_of_the_pectin_notebook(gpu_required, real, q1_route):
    """MBPLS(n_components=3).fit([x1, x2], y): every attribute the notebook reads (block importances A_ / A_corrected_,
    block scores T_, loadings P_, the scalers it inverts the loadings with) and everything else"""
    import warnings
    import mbpls_amd
    from mbpls_amd import MBPLS
    X1, X2, Y = real["x1"], real["x2"], real["y"].reshape(-1, 1)
    mbpls_amd.get_context().trim_pool()
    with warnings.catch_warnings():
        # (the spectra are offset by many standard deviations: the first queue of such a fit gives up, is queued again
        # with centring passes -- and normalised the zeros it had copied: 0 / 0, a RuntimeWarning of every first fit)
        warnings.simplefilter("error", RuntimeWarning)
        m = MBPLS(n_components=3).fit([X1, X2], real["y"])           # (1-D y, as in the notebook)
    w = util.compare(_flat_with_new(m, [X1, X2], Y), util.sub(real, "ref/two_blocks"), TOL, what="pectin, two blocks")
    assert w < 1e-9, w
    # "inverse transform loadings if standardize option was used during fitting" (the notebook's plotting cell)
    for b in range(2):
        ref_mean, ref_scale = real[f"ref/two_blocks/x_mean/{b}"], real[f"ref/two_blocks/x_scale/{b}"]
        for k in range(3):
            back = m.x_scalers_[b].inverse_transform(m.P_[b].T[k:k + 1])
            want = real[f"ref/two_blocks/P_/{b}"][:, k] * ref_scale + ref_mean
            s = np.sign(np.sum(real["ref/two_blocks/Ts_"][:, k] * m.Ts_[:, k]))
            mine = (s * m.P_[b][:, k]) * ref_scale + ref_mean
            assert util.relerr(np.ravel(back), m.P_[b][:, k] * ref_scale + ref_mean) < 1e-12
            assert util.relerr(mine, want) < 1e-9


def test_data_frames_go_in_as_in_the_notebooks(gpu_required, real):
    pd = pytest.importorskip("pandas")
    from mbpls_amd import MBPLS
    x1 = pd.DataFrame(real["x1"], index=real["y"], columns=real["wavenumbers"])
    x2 = pd.DataFrame(real["x2"], index=real["y"])
    m = MBPLS(n_components=3).fit([x1, x2], pd.DataFrame(real["y"]))
    w = util.compare(util.model_to_flat(m), util.sub(real, "ref/two_blocks"), TOL,
                     skip=("predict", "T_new", "Ts_new", "U_new"), what="pectin, data frames")
    assert w < 1e-9, w
    assert util.relerr(m.predict([x1, x2]), real["ref/two_blocks/predict"]) < 1e-9
    x12 = pd.concat((x1, x2), axis=1)
    one = MBPLS(n_components=3).fit(x12, real["y"])
    assert one.beta_.shape == (178, 1)


@pytest.mark.parametrize("name,key", [("ftir", "x1"), ("carb", "x2")])
def test_single_block_pls_of_either_block(gpu_required, real, name, key):
    from mbpls_amd import MBPLS
    X, Y = real[key], real["y"].reshape(-1, 1)
    m = MBPLS(n_components=5).fit(X, Y)
    w = util.compare(_flat_with_new(m, X, Y), util.sub(real, f"ref/{name}"), TOL, what=f"pectin, {name}")
    assert w < 1e-7, w
    # "mbpls.fit_transform(ftir, yield_all)" as the FTIR / microarray notebooks call it: a data frame and the yield
    pd = pytest.importorskip("pandas")
    m2 = MBPLS(n_components=5)
    Ts, U = m2.fit_transform(pd.DataFrame(X, index=real["y"]), real["y"])
    s = np.sign(np.sum(real[f"ref/{name}/Ts_"] * m2.Ts_, axis=0))
    assert util.relerr(Ts * s, real[f"ref/{name}/Ts_new"]) < 1e-7
    assert util.relerr(U * s, real[f"ref/{name}/U_new"]) < 1e-7
    assert util.relerr(m2.beta_[:, 0], real[f"ref/{name}/beta_"][:, 0]) < 1e-7      # the notebooks' bar plot


def test_leave_one_out_curve_over_the_number_of_components(gpu_required, real):
    """"Find the right number of LVs": cross_val_predict(MBPLS(k), [x1 | x2], y, cv=len(y)) for k = 1 .. 15 -- the folds
    batched per pass over the resident matrix (mbpls_amd.cross_val_predict), the fold-by-fold loop, and sklearn's own
    cross_val_predict driving the estimator as the notebooks do; same predictions, same MSE curve as the reference."""
    from sklearn.model_selection import cross_val_predict as sk_cvp
    from mbpls_amd import MBPLS, cross_val_predict
    X12, y = np.hstack((real["x1"], real["x2"])), real["y"]
    want = real["loo/x1x2"]
    mse_ref = ((want - y[:, None]) ** 2).mean(axis=0)
    worst = 0.0
    for k in range(1, 16):
        got = np.ravel(cross_val_predict(MBPLS(n_components=k), X12, y, cv=len(y)))
        e = util.relerr(got, want[:, k - 1])
        assert e < TOL, (k, e)
        worst = max(worst, e)
        assert abs(((got - y) ** 2).mean() - mse_ref[k - 1]) <= 1e-6 * mse_ref[k - 1]
    assert worst < 1e-7, worst
    for k in (3, 10):
        loop = np.ravel(cross_val_predict(MBPLS(n_components=k), X12, y, cv=len(y), batch_folds=False))
        assert util.relerr(loop, want[:, k - 1]) < 1e-7, k
        sk = np.ravel(sk_cvp(MBPLS(n_components=k), X12, y, cv=len(y)))
        assert util.relerr(sk, want[:, k - 1]) < 1e-7, k
    assert int(np.argmin(mse_ref)) == int(np.argmin([((np.ravel(cross_val_predict(MBPLS(n_components=k), X12, y,
                                                                                    cv=len(y))) - y) ** 2).mean()
                                                      for k in range(1, 16)]))


@pytest.mark.parametrize("name,key", [("ftir", "x1"), ("carb", "x2")])
def test_leave_one_out_of_either_block(gpu_required, real, name, key):
    from mbpls_amd import MBPLS, cross_val_predict
    X, y = real[key], real["y"]
    for c, k in enumerate((1, 3, 5)):
        got = np.ravel(cross_val_predict(MBPLS(n_components=k), X, y, cv=len(y)))
        assert util.relerr(got, real[f"loo/{name}"][:, c]) < 1e-7, (name, k)


def _flat_model(m, Xnew, Ynew, block_scores):
    flat = util.model_to_flat(m)
    flat["predict"] = m.predict(Xnew)
    if block_scores:
        Ts, T, U = m.transform(Xnew, Ynew, return_block_scores=True)
        for b, blk in enumerate(T):
            flat[f"T_new/{b}"] = blk
    else:
        Ts, U = m.transform(Xnew, Ynew)
    flat["Ts_new"], flat["U_new"] = Ts, U
    return flat


@pytest.mark.parametrize("layout", ["Xin2", "Xin4"])
def test_mbdata_mat(gpu_required, layout, kernel_route):
    """mbpls/data/MBdata.mat (fixture mbdata): ten samples x 100 variables in two or four blocks, three responses -- every
    method, standardised and not -- and the same matrix with four missing values through sparse_data=True"""
    from mbpls_amd import MBPLS
    g = util.load("mbdata")
    X, Y, Xmv = g["X"], g["Y"], g["Xmv"]
    blocks = lambda A: [np.ascontiguousarray(A[:, lo - 1:hi]) for lo, hi in g[layout]]
    worst = 0.0
    for method, full in (("NIPALS", False), ("UNIPALS", True), ("KERNEL", True), ("SIMPLS", True)):
        for std in (True, False):
            ref = util.sub(g, f"ref/{layout}/{method}/std{int(std)}")
            m = MBPLS(n_components=3, method=method, standardize=std, full_svd=full).fit(blocks(X), Y)
            has_blocks = any(k.startswith("T_new") for k in ref)
            worst = max(worst, util.compare(_flat_model(m, blocks(X), Y, has_blocks), ref, TOL,
                                            what=f"mbdata {layout} {method} std{std}"))
    assert worst < 1e-8, worst
    for std in (True, False):
        m = MBPLS(n_components=2, sparse_data=True, standardize=std).fit(blocks(Xmv), Y)
        w = util.compare(_flat_model(m, blocks(Xmv), Y, True), util.sub(g, f"sparse/{layout}/std{int(std)}"), TOL,
                         what=f"mbdata sparse {layout} std{std}")
        assert w < 1e-8, w


def test_everyday_sklearn_usage(gpu_required, real):
    """What users do around the estimator, with the reference's numbers for the same calls (fixture keys usage/*): score /
    r2_score / explained_variance_score, integer input, 1-D / list / Series responses, a Pipeline behind a StandardScaler,
    GridSearchCV over n_components, transform without Y, a pickle round trip, one new sample."""
    import pickle
    pd = pytest.importorskip("pandas")
    from sklearn.model_selection import GridSearchCV
    from sklearn.pipeline import Pipeline
    from sklearn.preprocessing import StandardScaler
    from mbpls_amd import MBPLS
    X1, X2, y = real["x1"], real["x2"], real["y"]
    m = MBPLS(n_components=3).fit([X1, X2], y)
    assert abs(m.score([X1, X2], y) - float(real["usage/score"])) < 1e-10
    assert abs(m.r2_score([X1, X2], y) - float(real["usage/r2_score"])) < 1e-10
    assert abs(m.explained_variance_score([X1, X2], y) - float(real["usage/explained_variance_score"])) < 1e-10
    Ts = m.transform([X1, X2])                                      # without Y: the super scores alone
    s = np.sign(np.sum(real["ref/two_blocks/Ts_"] * m.Ts_, axis=0))
    assert util.relerr(np.asarray(Ts) * s, real["usage/transform_without_y"]) < 1e-9
    assert m.predict([X1[:1], X2[:1]]).shape == (1, 1)
    m2 = pickle.loads(pickle.dumps(m))
    assert np.array_equal(m2.predict([X1, X2]), m.predict([X1, X2]))
    mi = MBPLS(n_components=2).fit((X1 * 1000).astype(int), y)
    assert mi.beta_.dtype == np.float64 and util.relerr(mi.beta_, real["usage/int_beta"]) < 1e-9
    for yy in (list(y), pd.Series(y), y.reshape(-1, 1)):
        assert MBPLS(n_components=2).fit(X1, yy).beta_.shape == (148, 1)
    pipe = Pipeline([("sc", StandardScaler()), ("pls", MBPLS(n_components=2))]).fit(X1, y)
    assert util.relerr(np.ravel(pipe.predict(X1)), real["usage/pipeline_predict"]) < 1e-9
    gs = GridSearchCV(MBPLS(), {"n_components": [1, 2, 3, 4]}, cv=5, scoring="neg_mean_squared_error").fit(X1, y)
    assert gs.best_params_["n_components"] == int(real["usage/grid_best_n_components"])
    assert util.relerr(np.asarray(gs.cv_results_["mean_test_score"]), real["usage/grid_mean_test_score"]) < 1e-8
