mkdir -p gpurun_out/r02f
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_rccl.py -x -q -m gpu > gpurun_out/r02f/full_rccl.log 2>&1; echo full_rccl rc=$?; tail -4 gpurun_out/r02f/full_rccl.log
python -m pytest tests/test_gpu_parity.py tests/test_gpu_sparse.py tests/test_gpu_long.py -x -q -m gpu -k "kernel or KERNEL or offset or missing or ngtp or long or matrix" > gpurun_out/r02f/sel.log 2>&1; echo sel rc=$?; tail -4 gpurun_out/r02f/sel.log
python bench.py --dist --p 250000 --steps 3 --warmup 1 --cpu-cols 0 --e2e-cols 0 > gpurun_out/r02f/bench_dist1.log 2>&1; echo dist1 rc=$?; tail -2 gpurun_out/r02f/bench_dist1.log | cut -c1-1500
MBPLS_BENCH_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 > gpurun_out/r02f/bench_shared2.log 2>&1; echo shared2 rc=$?; tail -3 gpurun_out/r02f/bench_shared2.log | cut -c1-2500
python tools/bench_configs.py c4 > gpurun_out/r02f/c4.log 2>&1; echo c4 rc=$?; tail -1 gpurun_out/r02f/c4.log | cut -c1-1500
MBPLS_IMPLICIT_STD=0 python tools/bench_configs.py c4 > gpurun_out/r02f/c4_stored.log 2>&1; tail -1 gpurun_out/r02f/c4_stored.log | cut -c1-1500
