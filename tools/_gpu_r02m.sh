R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02m
mkdir -p $O
python tools/outlier_probe.py default > $O/probe_default.txt 2>&1; cat $O/probe_default.txt
python bench.py --steps 9 --warmup 2 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_quick.log 2>&1; tail -1 $O/bench_quick.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['fit_wall_s'], d['fit_wall_s_each']); print(d['host_phases_ms_last_fit']); print(d['kernels'])"
