#!/bin/bash
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q -k "several or SIMPLS or simpls or q10 or responses or golden or random or medium" 2>&1 | tail -3
timeout 600 python3 tools/paper_protocol.py a --data planted --methods NIPALS,SIMPLS 2>/dev/null | grep '^{' | cut -c1-250
timeout 600 python3 tools/paper_protocol.py b --data planted --methods NIPALS,SIMPLS 2>/dev/null | grep '^{' | cut -c1-250
for c in simpls nipals_q10 unipals_q10; do timeout 600 python3 tools/bench_configs.py $c 2>/dev/null | tail -1 | cut -c1-200; done
