"""rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of tools/long_one.py -> one JSON:
HBM bytes per launch of every cluster pass (csrc/colpass_cluster.hip) against the algorithmic 8np.

    python tools/pmc_long_to_json.py DIR n p OUT.json     (DIR/FETCH_SIZE/..., DIR/WRITE_SIZE/...)

Unit / correction as MI355X_MICROARCH.md section HBM prescribes: derived counters in KiB; on gfx950 FETCH_SIZE tallies the
128-byte requests of a wide coalesced (16 B/lane) stream as 64 bytes -> read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x
1024 is exact."""
import collections
import csv
import glob
import json
import os
import sys

root, n, p, outp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
# k_colpass_cl<R, TMAX, D, L, DED, STATS, DEFL, SWEEP, CEN>
out = {"config": {"n": n, "p": p, "K": 3, "method": "NIPALS", "kernels": "clusters of workgroups (one read per pass)"},
       "method": __doc__.split("Unit / correction")[1].strip(), "algorithmic_bytes_per_pass": 8.0 * n * p, "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(files[0])):
        kn = r["Kernel_Name"]
        if "k_colpass_cl<" not in kn:
            continue
        a = [x.strip() for x in kn.split("<", 1)[1].split(">")[0].split(",")]
        stats, defl, sweep = a[5] == "true", a[6] == "true", a[7] == "true"
        nm = "standardize_sweep" if stats else ("deflate_sweep" if defl and sweep else ("deflate" if defl else "sweep"))
        agg[nm].append(float(r["Counter_Value"]))
    for nm, v in agg.items():
        d = out["kernels"].setdefault(nm, {})
        d[c + "_KiB_avg"] = sum(v) / len(v)
        d["launches"] = len(v)
for nm, d in out["kernels"].items():
    d["hbm_read_bytes"] = 2 * d["FETCH_SIZE_KiB_avg"] * 1024
    d["hbm_write_bytes"] = d["WRITE_SIZE_KiB_avg"] * 1024
    d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    d["traffic_over_algorithmic"] = d["hbm_bytes_per_launch"] / (8.0 * n * p)
    print(f"{nm:18s} read {d['hbm_read_bytes'] / 1e9:7.3f} GB  write {d['hbm_write_bytes'] / 1e9:7.3f} GB  "
          f"= {d['traffic_over_algorithmic']:.3f} x 8np")
json.dump(out, open(outp, "w"), indent=1)
