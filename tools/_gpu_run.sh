cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "cross_val or row_subset" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_interop.py -x -q 2>&1 | tail -3
timeout 900 python tools/time_cv.py 2>&1 | tail -16
