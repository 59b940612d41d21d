"""sparse_data=True far beyond the register shapes: n = 1 000 003 samples, 3 % missing, against the CPU oracle (a one-off check;
the suite tests up to 131 072 samples).   python tools/sparse_million_rows.py"""
import warnings, numpy as np, sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from oracle import mbpls_oracle as orc, mbpls_oracle_sparse as ors
from tests import util
from mbpls_amd import MBPLS
n, p, q = 1_000_003, 24, 1
X, Y = orc.make_latent(n, p, q, r=4, seed=5)
rng = np.random.RandomState(2); X[rng.rand(n, p) < 0.03] = np.nan
blocks = [X[:, :9], X[:, 9:]]
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    t = time.time(); ref = ors.fit(blocks, Y, n_components=3); t_or = time.time() - t
    mdl = MBPLS(n_components=3, sparse_data=True).fit(blocks, Y)
    t = time.time(); mdl = MBPLS(n_components=3, sparse_data=True).fit(blocks, Y); t_gpu = time.time() - t
w = util.compare(util.model_to_flat(mdl), util.model_to_flat(ref), 1e-6, what="sparse n=1e6")
Xn = [b[:100] for b in blocks]
print(f"sparse_data=True n={n} p={p}: worst rel. deviation from the oracle {w:.2e}; predict {util.relerr(mdl.predict(Xn), ors.predict(ref, Xn)):.2e}; "
      f"iterations {mdl.nipals_iterations_} / {ref.nipals_iterations_}; oracle {t_or:.1f} s, fit from host arrays {1e3 * t_gpu:.0f} ms")
