cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "ragged_sample" 2>&1 | tail -6
timeout 600 python tools/ab_rows.py 4500,5000 2>&1 | grep "^n=\|failed"
