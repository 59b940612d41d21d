"""cProfile of one batched leave-one-out run (cv_batch): python tools/profile_cv.py [n] [p] [K]"""
import cProfile, os, pstats, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
warnings.simplefilter("ignore")
from mbpls_amd import MBPLS, cross_val_predict
from oracle import mbpls_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
p = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
K = int(sys.argv[3]) if len(sys.argv) > 3 else 5
X, Y = orc.make_latent(n, p, 1, seed=1)
m = MBPLS(n_components=K)
for _ in range(2):
    cross_val_predict(m, X, Y, cv=n)
pr = cProfile.Profile(); pr.enable()
cross_val_predict(m, X, Y, cv=n)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
