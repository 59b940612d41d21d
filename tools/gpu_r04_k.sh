#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r04ev
mkdir -p $O
( timeout 900 python3 tools/paper_protocol.py a --data both 2>/dev/null; timeout 1500 python3 tools/paper_protocol.py b --data both 2>/dev/null ) | grep '^{' > $O/r04_paper_protocol.jsonl
wc -l $O/r04_paper_protocol.jsonl
bash tools/gpu_suite.sh 2>&1 | tail -5
