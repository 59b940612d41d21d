cd $GRAFT_REPO_ROOT
for d in 2 3 1 2 3; do echo "stats prefetch $d"; MBPLS_HIP_STATS_PREFETCH=$d python bench.py --steps 6 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras 2>/dev/null | grep -a '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['fit_wall_s'], d['kernels']['standardize_sweep'], d['kernels']['sweep']['avg_ms'])"; done
