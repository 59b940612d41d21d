cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q 2>&1 | tail -6
