#!/bin/bash
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -x -q -k "UNIPALS or unipals or ngtp or golden or random" 2>&1 | tail -3
timeout 600 python3 tools/paper_protocol.py a --data planted --methods UNIPALS 2>/dev/null | grep '^{' | cut -c1-260
timeout 600 python3 tools/paper_protocol.py b --data planted --methods UNIPALS 2>/dev/null | grep '^{' | cut -c1-260
