#!/bin/bash
timeout 600 python3 tools/try_wake.py 2>&1 | tail -13
