#!/bin/bash
timeout 900 python3 -m pytest tests/test_gpu_few_samples.py tests/test_gpu_long.py tests/test_gpu_fuzz.py -m gpu -x -q 2>&1 | tail -5
