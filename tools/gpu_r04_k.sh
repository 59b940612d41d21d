#!/bin/bash
timeout 900 python3 -m pytest tests/test_gpu_small_kernels.py -m gpu -x -q -k "r_update or transpose" 2>&1 | tail -6
