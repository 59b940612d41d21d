#!/bin/bash
MBPLS_COPIER_TRACE=1 timeout 300 python3 tools/host_timeline.py --rccl --n 4096 --p 125000 --K 10 --fits 8 --fresh --show last 2>&1 | grep "copier\] job" | tail -13 | cut -c1-200
for i in 1 2; do timeout 300 python3 tools/shard_fixed.py 2>&1 | grep workload | cut -c1-330; done
timeout 300 python3 tools/gap_fit.py 1000000
