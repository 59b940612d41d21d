#!/bin/bash
python3 - <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
from tests import fuzz
tot_bad = 0
for seed in (101, 202, 303, 404):
    bad, worst = fuzz.run(200, seed=seed, verbose=False)
    print("small seed", seed, "bad", bad, "worst", worst, flush=True); tot_bad += bad
for seed in (7, 8):
    bad, worst = fuzz.run(30, seed=seed, verbose=False, big=True)
    print("big seed", seed, "bad", bad, "worst", worst, flush=True); tot_bad += bad
print("TOTAL BAD", tot_bad)
PY
