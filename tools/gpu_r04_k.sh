#!/bin/bash
mkdir -p gpurun_out/r04k
timeout 300 python3 tools/host_timeline.py --q 10 --blocks 2 --n 20000 --p 1000 --K 20 --fits 4 --fresh --show last --min-us 15 > gpurun_out/r04k/host_timeline_paper_a.txt 2>&1
head -2 gpurun_out/r04k/host_timeline_paper_a.txt | cut -c1-400; sed -n 3,400p gpurun_out/r04k/host_timeline_paper_a.txt | head -90
