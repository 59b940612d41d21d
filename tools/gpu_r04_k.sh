#!/bin/bash
timeout 300 python3 tools/profile_fit.py --n 20000 --p 1000 --q 10 --K 20 --blocks 2 2>&1 | grep -v "^$" | head -60
