cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import cProfile, pstats, sys, os, warnings, time
sys.path.insert(0, os.getcwd())
import numpy as np
warnings.simplefilter("ignore")
from mbpls_amd import MBPLS, cross_val_predict
from oracle import mbpls_oracle as orc
n,p,K,q=150,3000,5,3
X,Y=orc.make_latent(n,p,q,seed=1)
Xs=[np.ascontiguousarray(X[:,:p//2]),np.ascontiguousarray(X[:,p//2:])]
m=MBPLS(n_components=K)
cross_val_predict(m,Xs,Y,cv=20)
t0=time.perf_counter(); cross_val_predict(m,Xs,Y,cv=n); print("total", time.perf_counter()-t0)
pr=cProfile.Profile(); pr.enable(); cross_val_predict(m,Xs,Y,cv=n); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
PY
