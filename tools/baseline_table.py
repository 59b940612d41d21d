"""BASELINE.md section 4 (the paper's protocol on one MI355X) from profiles/r04_paper_protocol.jsonl, with round 3's
fit times and CPU-oracle column from profiles/r03_paper_protocol.jsonl.   python tools/baseline_table.py  (rewrites the section)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, "BASELINE.md")
s = open(p).read()
a = s.index("## 4. The paper's protocol on one MI355X")
head_end = s.index("| shape | data | method |", a)
tail = s.index("\nUNIPALS at shape (b)", a)
cpu, r3 = {}, {}
for ln in open(os.path.join(ROOT, "profiles", "r03_paper_protocol.jsonl")):
    d = json.loads(ln)
    key = (d["shape"][0], d["data"], d["method"])
    c = d.get("cpu_oracle") or {}
    if "s_per_iteration" in c:
        cpu[key] = (f"{c['s_per_iteration']:.4f} s per iteration × "
                    f"{d.get('iterations_total_earlier_n_space_loop', d.get('iterations_total'))} iterations = "
                    f"{c.get('s_for_the_iterations_of_the_gpu_fit', '?')} s")
    elif "fit_s" in c:
        cpu[key] = f"{c['fit_s']} s"
    elif "skipped" in c:
        cpu[key] = "not run: " + c["skipped"]
    if "fit_s_from_host_arrays" in d:
        r3[key] = min(d["fit_s_from_host_arrays"])
rows = ["| shape | data | method | paper, 2× Xeon X5650 (read off the figure) | this build: `fit()` from host arrays, best of 3 | "
        "same, X resident in HBM | round 3, from host arrays | CPU oracle, faithful mode, GPU box host (16 cores) | "
        "NIPALS iterations (cap hits) |", "|---|---|---|---|---|---|---|---|---|"]
for ln in open(os.path.join(ROOT, "profiles", "r04_paper_protocol.jsonl")):
    d = json.loads(ln)
    key = (d["shape"][0], d["data"], d["method"])
    paper = d.get("paper_fit_s_2x_Xeon_X5650")
    paper_s = f"≈ {paper} s" if paper else "— (curve leaves the plot at N = P ≈ 6000, > 400 s)"
    host, dev = min(d["fit_s_from_host_arrays"]), min(d["fit_s_resident"])
    it = f"{d['iterations_total']} ({d.get('cap_hits', 0)})" if "nipals_iterations" in d else ""
    prev = f"{r3[key]:.4f} s" if key in r3 else "not run"
    c = cpu.get(key, "not run (the reference's dense D^T VAR D: 2 × 8e12 flop per component)"
                if d["method"] == "KERNEL" and key[0] == "b" else "")
    rows.append(f"| {key[0]} | {key[1]} | {key[2]} | {paper_s} | **{host:.4f} s** | {dev:.4f} s | {prev} | {c} | {it} |")
open(p, "w").write(s[:head_end] + "\n".join(rows) + "\n" + s[tail:])
print("\n".join(rows))
