#!/bin/bash
mkdir -p gpurun_out/r04k
R=$GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_copier.py tests/test_gpu_parity.py tests/test_gpu_rccl.py -m gpu -x -q -k "copier or golden or c2 or column_by_column or fixed_point or unipals or rccl or ranks" 2>&1 | tail -4
for i in 1 2; do timeout 300 python3 tools/shard_fixed.py --local 2>&1 | tail -1 | cut -c1-250; done
timeout 300 python3 tools/shard_fixed.py 2>&1 | grep workload | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gaps; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/gaps -- python3 $R/tools/gap_fit.py 125000 rccl > /tmp/gaps.log 2>&1
python3 $R/tools/trace_gaps.py /tmp/gaps --list > $R/gpurun_out/r04k/gaps_p125k_rccl.txt 2>&1; head -1 $R/gpurun_out/r04k/gaps_p125k_rccl.txt; grep "k_step_reg" $R/gpurun_out/r04k/gaps_p125k_rccl.txt | head -8; tail -10 $R/gpurun_out/r04k/gaps_p125k_rccl.txt
