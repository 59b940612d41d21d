"""Build libmbpls_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: one .o per .hip, one link.

    python -m mbpls_amd.build [--force] [--save-temps]
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmbpls_hip.so")
SOURCES = ["ctx.hip", "colpass.hip", "colpass_nan.hip", "colpass_nan_long.hip", "colpass_long.hip", "colpass_cluster.hip", "colpass_sub.hip", "gemv.hip", "gram.hip", "gen.hip", "small.hip", "xtv_mfma.hip", "xw_mfma.hip", "nipals_step.hip", "axpy_cols.hip", "resident.hip", "step_reg.hip", "cv_batch.hip", "rccl_comm.hip", "copier.hip"]
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
         "-ffp-contract=off"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, save_temps=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "colpass_util.h"), os.path.join(HERE, "..", "include", "mbpls_hip.h")]
    hipcc = _hipcc()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if save_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append((s, cmd))
    def run(job):
        name, cmd = job
        r = subprocess.run(cmd, capture_output=True, text=True)
        return name, r
    with cf.ThreadPoolExecutor(max_workers=min(4, max(1, len(jobs)))) as ex:
        for name, r in ex.map(run, jobs):
            if verbose and (r.stderr.strip() or r.returncode):
                sys.stderr.write(f"--- {name}\n{r.stderr}\n")
            if r.returncode:
                raise RuntimeError(f"hipcc failed on {name}")
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            sys.stderr.write(r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--save-temps", action="store_true")
    a = ap.parse_args()
    print(build(force=a.force, save_temps=a.save_temps))
