cd $GRAFT_REPO_ROOT
for c in nipals_q10 unipals_q10 simpls c3; do
  timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print(d['config'], d['fit_s'], {k: (v['launches'], v['avg_ms']) for k, v in d['kernels'].items()}, d['host_phases_ms'])"
done
