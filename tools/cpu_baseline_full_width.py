"""the CPU baseline MEASURED at the full width of BASELINE config[1] -- once, on the GPU box's host.

The oracle in faithful mode (the reference's numpy op sequence, n x p temporaries included) fits the whole
n = 4096 x p = 1,000,000 matrix bench.py times the GPU on (generated in HBM, downloaded: 32.8 GB; the faithful fit
peaks near 5 x that in host RAM and takes minutes).  bench.py's default run keeps its bounded samples and labelled
extrapolation; it reads the JSON this tool leaves (profiles/r06_cpu_baseline_full_width.json) and reports the measured
figure beside the extrapolated one.  Bonus: the GPU fit of the SAME million columns is compared with that oracle run
attribute by attribute -- parity at BASELINE size against the CPU restatement itself, not through properties.
    python tools/cpu_baseline_full_width.py [p] [lean]       # one JSON line on stdout"""
import json
import os
import platform
import resource
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import mbpls_amd
from bench import planted
from mbpls_amd import MBPLS, DeviceMatrix
from mbpls_amd.runtime import cpu_quota
from oracle import mbpls_oracle as orc
from tests import util

n, K = 4096, 10
p = next((int(v) for v in sys.argv[1:] if v.isdigit()), 1_000_000)
with open("/proc/meminfo") as f:
    avail = next(int(line.split()[1]) * 1024 for line in f if line.startswith("MemAvailable"))
need = 6.5 * 8.0 * n * p
out = {"n": n, "p": p, "K": K, "method": "NIPALS", "host_mem_available_GB": round(avail / 1e9, 1),
       "host_mem_needed_GB": round(need / 1e9, 1), "date": time.strftime("%Y-%m-%d"), "host": platform.node(),
       "cpus_visible": os.cpu_count()}
if avail < need:
    out["error"] = "not enough host memory for the faithful oracle at this width"
    print(json.dumps(out)); sys.exit(1)
ctx = mbpls_amd.get_context()
Ts, Y = planted(n, 1, 32, seed=0)
X = DeviceMatrix(ctx, n, p, [0, p])
dT = ctx.from_host(Ts)
X.fill_latent(dT, 32, 0.5, 1234, 0)
ctx.sync()
t0 = time.perf_counter()
g = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
t0 = time.perf_counter()
g = MBPLS(n_components=K, method="NIPALS").fit(X, Y)
out["gpu_fit_s"] = round(time.perf_counter() - t0, 4)
t0 = time.perf_counter()
Xh = X.download()
out["download_s"] = round(time.perf_counter() - t0, 2)
X.close()
thr = cpu_quota()
out["blas_threads"] = thr
import threadpoolctl
with threadpoolctl.threadpool_limits(limits=thr):
    orc.fit(np.ascontiguousarray(Xh[:, :8192]), Y, n_components=K, method="NIPALS", faithful=True)      # warm-up
    for mode, faithful in (("faithful", True),) + ((("lean", False),) if "lean" in sys.argv else ()):
        t0 = time.perf_counter()
        ref = orc.fit(Xh, Y, n_components=K, method="NIPALS", standardize=True, calc_all=True, faithful=faithful)
        out[mode + "_fit_s"] = round(time.perf_counter() - t0, 2)
        out[mode + "_iterations"] = list(ref.nipals_iterations_)
        w = util.compare(util.model_to_flat(g), util.model_to_flat(ref), 1e-6, what=f"GPU vs {mode} oracle at full width")
        out["gpu_vs_" + mode + "_oracle_worst_relerr_all_attributes"] = w
out["peak_rss_GB"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024 / 1e9, 1)
out["measured_full_width_s"] = out["faithful_fit_s"]
out["gpu_speedup_fits_per_s"] = round(out["faithful_fit_s"] / out["gpu_fit_s"], 1)
print(json.dumps(out))
