"""Timeline of the LAST fit in a rocprofv3 --kernel-trace CSV (tools/gap_fit.py under the profiler): every kernel in
launch order with its duration and the idle gap in front of it, summed per kernel name.

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -- python3 $REPO/tools/gap_fit.py 125000
    python tools/trace_gaps.py OUT [--list] [--split-ms 100] [--pick last|slowest]
"""
import csv
import glob
import os
import re
import sys

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
assert files, f"no kernel trace under {d}"
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void mbpls::", "").replace("mbpls::", "")
    m = re.match(r"(\w+)<(.*)>", name)
    return name if not m else f"{m.group(1)}<{m.group(2)[:40]}>"


def opt(name, default):
    return sys.argv[sys.argv.index(name) + 1] if name in sys.argv else default


# fits are separated by >= 100 ms of idle (gap_fit.py sleeps 0.2 s between them)
split_ns = float(opt("--split-ms", "100")) * 1e6
fits, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > split_ns:
        fits.append(cur)
        cur = []
    cur.append(b)
fits.append(cur)
fit = fits[-1]
if opt("--pick", "last") == "slowest":
    fit = max((f for f in fits if len(f) > 8), key=lambda f: f[-1][1] - f[0][0])
    print("spans of all bursts (ms):", [round((f[-1][1] - f[0][0]) / 1e6, 2) for f in fits if len(f) > 8])
span = (fit[-1][1] - fit[0][0]) / 1e6
busy = sum(e - s for s, e, _ in fit) / 1e6
print(f"{len(fits)} bursts; last: {len(fit)} kernels, first start -> last end {span:.3f} ms, busy {busy:.3f} ms, idle {span - busy:.3f} ms")
agg = {}
prev_end = fit[0][0]
for s, e, nm in fit:
    k = short(nm)
    a = agg.setdefault(k, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += max(0, s - prev_end) / 1e3
    if "--list" in sys.argv:
        print(f"  +{max(0, s - prev_end) / 1e3:8.1f} us gap | {(e - s) / 1e3:9.1f} us  {k}")
    prev_end = max(prev_end, e)
print(f"{'kernel':60s} {'n':>4s} {'sum us':>10s} {'avg us':>9s} {'gap before, sum us':>19s}")
for k, (c, du, gp) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:60s} {c:4d} {du:10.1f} {du / c:9.1f} {gp:19.1f}")
