"""One Gram launch (after a warm-up) for counter collection: python tools/gram_one.py [n] [p]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbpls_amd
from mbpls_amd import DeviceMatrix
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
p = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
ctx = mbpls_amd.get_context()
X = DeviceMatrix(ctx, n, p)
X.fill_latent(ctx.from_host(np.random.RandomState(0).randn(n, 8)), 8, 0.5, 1, 0)
G = ctx.empty((n, n))
X.gram(G, n); ctx.sync()
X.gram(G, n); ctx.sync()
