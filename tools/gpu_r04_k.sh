#!/bin/bash
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "column_by_column" 2>&1 | tail -12
