cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_few_samples.py -x -q -k "ragged or medium_vs_oracle or golden or quickstart or few or offset or column_pass or generic" 2>&1 | tail -3
