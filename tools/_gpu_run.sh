cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "not fullsize and not many_components and not exhausted" 2>&1 | tail -5
for c in simpls nipals_q10 unipals_q10; do timeout 600 python tools/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-330; done
