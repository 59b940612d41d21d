mkdir -p gpurun_out/r02k
python tools/outlier_probe.py default > gpurun_out/r02k/probe_default.txt 2>&1; cat gpurun_out/r02k/probe_default.txt
python tools/outlier_probe.py gc_off > gpurun_out/r02k/probe_gc_off.txt 2>&1; cat gpurun_out/r02k/probe_gc_off.txt
MBPLS_RESIDENT=0 python tools/outlier_probe.py default > gpurun_out/r02k/probe_host.txt 2>&1; cat gpurun_out/r02k/probe_host.txt
