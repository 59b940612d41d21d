#!/bin/bash
for i in 1 2; do
timeout 300 python3 tools/shard_fixed.py 2>&1 | grep workload | cut -c90-330
timeout 300 python3 tools/shard_fixed.py --with-big 2>&1 | grep workload | cut -c90-330
done
