"""cProfile of 200 warm small fits from host arrays (n=150, p=3000, two blocks): python tools/profile_small_fit.py [method]"""
import cProfile, os, pstats, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.simplefilter("ignore")
from mbpls_amd import MBPLS
from oracle import mbpls_oracle as orc
method = sys.argv[1] if len(sys.argv) > 1 else "NIPALS"
X, Y = orc.make_latent(150, 3000, 1, seed=1)
Xs = [np.ascontiguousarray(X[:, :1500]), np.ascontiguousarray(X[:, 1500:])]
m = MBPLS(n_components=5, method=method)
for _ in range(5):
    m.fit(Xs, Y)
pr = cProfile.Profile(); pr.enable()
for _ in range(200):
    m.fit(Xs, Y)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
