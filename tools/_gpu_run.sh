cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py --cpu-cols 0 --e2e-cols 0 > gpurun_out/bench_quick.log 2>&1; echo rc=$?
tail -1 gpurun_out/bench_quick.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['fit_wall_s']); print(json.dumps(d['other_solvers_same_matrix'], indent=1))"
