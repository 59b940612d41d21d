cd $GRAFT_REPO_ROOT
timeout 300 python tools/profile_cv.py 2>&1 | head -24; timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k cross_val 2>&1 | tail -2; timeout 600 python tools/time_cv.py 2>&1 | head -3
