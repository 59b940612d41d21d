cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "subspace or several or faithful or medium_vs_oracle or both_deflation" 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_long.py tests/test_gpu_few_samples.py -x -q 2>&1 | tail -4
timeout 300 python tools/host_profile_q.py 1000000 10 10 NIPALS 2 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl" | head -12
