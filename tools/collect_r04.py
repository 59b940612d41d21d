"""After `gpurun -- 'bash tools/gpu_evidence_r04.sh'`: copy the round-4 evidence from gpurun_out/r04ev into profiles/ and
print the BASELINE.md section-4 table rows from the paper-protocol lines.   python tools/collect_r04.py [--table]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r04ev")
DST = os.path.join(ROOT, "profiles")


def copy(src, name=None):
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(DST, name or os.path.basename(src)))
        print("copied", name or os.path.basename(src))
    else:
        print("MISSING", src)


for f in sorted(glob.glob(os.path.join(SRC, "r04_*"))):
    copy(f)
st = glob.glob(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    copy(st[0], "r04_bench_c2_rocprofv3_kernel_stats.csv")
log = os.path.join(SRC, "bench_under_rocprof.log")
if os.path.exists(log):
    lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{")]
    if lines:
        open(os.path.join(DST, "r04_bench_c2_line_under_rocprofv3.json"), "w").write(lines[-1] + "\n")
for tag in ("n16384", "n1000000"):
    st = glob.glob(os.path.join(SRC, f"long_stats_{tag}", "**", "*kernel_stats.csv"), recursive=True)
    if st:
        copy(st[0], f"r04_long_{tag}_rocprofv3_kernel_stats.csv")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        cc = glob.glob(os.path.join(SRC, f"long_pmc_{tag}", c, "**", "*counter_collection.csv"), recursive=True)
        if cc:
            copy(cc[0], f"r04_pmc_{c}_long_{tag}_counter_collection.csv")

if "--table" in sys.argv:
    path = os.path.join(DST, "r04_paper_protocol.jsonl")
    for ln in open(path):
        d = json.loads(ln)
        shape = d["shape"][0]
        paper = d.get("paper_fit_s_2x_Xeon_X5650")
        paper_s = f"≈ {paper} s" if paper else "— (curve leaves the plot at N=P≈6000, > 400 s)"
        host, dev = min(d["fit_s_from_host_arrays"]), min(d["fit_s_resident"])
        it = ""
        if "nipals_iterations" in d:
            it = f"{d['iterations_total']} ({d.get('cap_hits', 0)})"
        print(f"| {shape} | {d['data']} | {d['method']} | {paper_s} | **{host:.4f} s** | {dev:.4f} s | {it} | "
              f"{d['fit_plan'].get('kernels', '')} |")
