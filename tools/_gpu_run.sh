cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long.py tests/test_gpu_few_samples.py tests/test_gpu_rccl.py -x -q -k "not fullsize and not exhausted" 2>&1 | tail -5
for c in nipals_q10 unipals_q10; do timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-330; done
