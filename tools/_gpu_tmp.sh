cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py --cpu-cols 0 --e2e-cols 0 --no-extras 2>/dev/null | grep -a '^{"metric"' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['fit_wall_s'], d['fit_wall_s_each']); print(d['host_phases_ms_last_fit'])"
