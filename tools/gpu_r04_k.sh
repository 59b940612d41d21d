#!/bin/bash
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_fuzz.py tests/test_gpu_few_samples.py -m gpu -x -q 2>&1 | tail -4
timeout 600 python3 tools/paper_protocol.py b --data planted --methods KERNEL,UNIPALS 2>/dev/null | grep '^{' | cut -c1-330
