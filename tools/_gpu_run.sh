cd $GRAFT_REPO_ROOT
for c in nipals_q10 unipals_q10; do
  timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config'], d['fit_s'], d['host_phases_ms'])"
done
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "never_writes or several or faithful" 2>&1 | tail -3
