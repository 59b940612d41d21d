#!/bin/bash
for i in 1 2 3; do timeout 600 python3 -X faulthandler -m pytest tests/test_gpu_copier.py -m gpu -x -q 2>&1 | tail -1; done
timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_rccl.py -m gpu -x -q -k "golden or c2 or column_by_column or rccl" 2>&1 | tail -2
