cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_long.py tests/test_gpu_few_samples.py tests/test_gpu_interop.py tests/test_gpu_rccl.py tests/test_gpu_fuzz.py tests/test_gpu_sparse.py -x -q -k "not fullsize and not exhausted" 2>&1 | tail -12
timeout 300 python tools/host_profile_q.py 1000000 10 10 NIPALS 2 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | head -30
