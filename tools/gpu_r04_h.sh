cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
( timeout 600 python tools/paper_protocol.py a --data planted 2>/dev/null; timeout 900 python tools/paper_protocol.py b --data planted 2>/dev/null ) | grep '^{' > $O/paper_planted.jsonl
python3 - <<'PY'
import json
for ln in open('gpurun_out/r04h/paper_planted.jsonl'):
    d=json.loads(ln); print(d['shape'][0], d['method'], min(d['fit_s_from_host_arrays']), min(d['fit_s_resident']), d['fit_plan'].get('solver'))
PY
( timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "several_responses_nipals_never or q1_fixed_point or unipals" ) 2>&1 | tail -3
timeout 300 python3 bench.py --no-first-fit --cpu-cols 0 --e2e-cols 0 --steps 5 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['fit_wall_s'], d['optional_q1_fixed_point'], d['other_solvers_same_matrix'])"
