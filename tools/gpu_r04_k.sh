#!/bin/bash
mkdir -p gpurun_out/r04k
R=$GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "golden or c2 or fixed_point or exhausted or random or NIPALS" 2>&1 | tail -4
timeout 300 python3 tools/shard_fixed.py --local 2>&1 | tail -2 | cut -c1-250
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -- python3 $R/tools/gap_fit.py 125000 rccl > /tmp/gaps.log 2>&1; tail -3 /tmp/gaps.log
python3 $R/tools/trace_gaps.py /tmp/gaps --list > $R/gpurun_out/r04k/gaps_p125k_rccl.txt 2>&1; tail -12 $R/gpurun_out/r04k/gaps_p125k_rccl.txt; grep k_step_reg $R/gpurun_out/r04k/gaps_p125k_rccl.txt | head -22
