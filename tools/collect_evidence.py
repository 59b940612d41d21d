"""After `gpurun -- 'bash tools/gpu_evidence.sh r05'`: copy that round's evidence from gpurun_out/<round>ev into profiles/
and (--table) print the BASELINE.md section-4 rows from the paper-protocol lines.
    python tools/collect_evidence.py r05 [--table]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RND = next((a for a in sys.argv[1:] if not a.startswith("-")), "r05")
SRC = os.path.join(ROOT, "gpurun_out", RND + "ev")
DST = os.path.join(ROOT, "profiles")


def copy(src, name=None):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(DST, name or os.path.basename(src)))
        print("copied", name or os.path.basename(src))
    else:
        print("MISSING", src)


for f in sorted(glob.glob(os.path.join(SRC, RND + "_*"))):
    copy(f)
st = glob.glob(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    copy(st[0], f"{RND}_bench_c2_rocprofv3_kernel_stats.csv")
log = os.path.join(SRC, "bench_under_rocprof.log")
if os.path.exists(log):
    lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
    if lines:
        open(os.path.join(DST, f"{RND}_bench_c2_line_under_rocprofv3.json"), "w").write(lines[-1] + "\n")

if "--table" in sys.argv:
    for ln in open(os.path.join(DST, f"{RND}_paper_protocol.jsonl")):
        d = json.loads(ln)
        shape = d["shape"][0]
        paper = d.get("paper_fit_s_2x_Xeon_X5650")
        paper_s = f"≈ {paper} s" if paper else "— (curve leaves the plot at N=P≈6000, > 400 s)"
        host, dev = min(d["fit_s_from_host_arrays"]), min(d["fit_s_resident"])
        it = f"{d['iterations_total']} ({d.get('cap_hits', 0)})" if "nipals_iterations" in d else ""
        print(f"| {shape} | {d['data']} | {d['method']} | {paper_s} | **{host:.4f} s** | {dev:.4f} s | {it} | "
              f"{d['fit_plan'].get('kernels', '')} |")
