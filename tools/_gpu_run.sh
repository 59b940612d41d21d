cd $GRAFT_REPO_ROOT
for c in nipals_q10 unipals_q10; do
  timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('default      ', d['config'], d['fit_s'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
  MBPLS_KERNEL_IMPLICIT=0 timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('stored copy  ', d['config'], d['fit_s'], {k: v['avg_ms'] for k, v in d['kernels'].items()})"
done
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long.py tests/test_gpu_few_samples.py tests/test_gpu_interop.py -x -q -k "not fullsize and not exhausted" 2>&1 | tail -5
