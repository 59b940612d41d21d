"""Turn the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras` into profiles/<round>_pmc_traffic_c2.json.

    python tools/pmc_to_json.py gpurun_out/pmc  [n] [p] [round prefix, default r03]

Unit/correction as MI355X_MICROARCH.md section HBM prescribes: the derived counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide coalesced (16 B/lane) stream as 64 bytes, so
read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact for 16 B/lane streaming stores."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02q/pmc"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
p = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
rnd = sys.argv[4] if len(sys.argv) > 4 else "r03"
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# k_colpass<R, TMAX, D, STD, DEFL, SWEEP, FULL>
names = {("true", "false", "true"): "standardize_sweep", ("false", "false", "true"): "sweep",
         ("false", "true", "true"): "deflate_sweep", ("false", "true", "false"): "deflate"}
out = {"config": {"n": n, "p": p, "K": 10, "method": "NIPALS", "deflation": os.environ.get("MBPLS_DEFLATION", "lazy")},
       "method": __doc__.split("Unit/correction")[1].strip(), "kernels": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)[0]
    shutil.copy(f, os.path.join(here, "profiles", f"{rnd}_pmc_{c}_bench_c2_counter_collection.csv"))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        kn = r["Kernel_Name"]
        if "k_colpass<" not in kn:
            continue
        args = [a.strip() for a in kn.split("<", 1)[1].split(">")[0].split(",")]
        nm = names.get(tuple(args[3:6]))
        if nm:
            agg[nm].append(float(r["Counter_Value"]))
    for nm, v in agg.items():
        d = out["kernels"].setdefault(nm, {})
        d[c + "_KiB_avg"] = sum(v) / len(v)
        d["launches"] = len(v)
for nm, d in out["kernels"].items():
    d["hbm_read_bytes"] = 2 * d["FETCH_SIZE_KiB_avg"] * 1024
    d["hbm_write_bytes"] = d["WRITE_SIZE_KiB_avg"] * 1024
    d["hbm_bytes_per_launch"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
    print(f"{nm:18s} read {d['hbm_read_bytes'] / 1e9:7.2f} GB  write {d['hbm_write_bytes'] / 1e9:7.2f} GB  "
          f"= {d['hbm_bytes_per_launch'] / (8.0 * n * p):.3f} units of 8np")
json.dump(out, open(os.path.join(here, "profiles", f"{rnd}_pmc_traffic_c2.json"), "w"), indent=1)
