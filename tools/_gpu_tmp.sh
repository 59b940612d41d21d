cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_sparse.py -x -q -m gpu -k "missing_responses_medium" 2>&1 | grep -E "^E |passed|failed" | head -12
