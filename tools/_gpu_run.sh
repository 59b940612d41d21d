cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_small_kernels.py tests/test_gpu_interop.py tests/test_gpu_few_samples.py -x -q 2>&1 | tail -5
