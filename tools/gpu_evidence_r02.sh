R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02q
mkdir -p $O
cd $R
python bench.py > $O/bench_full.log 2>&1; echo bench rc=$?; tail -1 $O/bench_full.log > $O/r02_bench_c2_line.json
python bench.py --dist --p 250000 --steps 3 --warmup 2 --cpu-cols 0 --e2e-cols 0 > $O/bench_dist1.log 2>&1; echo dist1 rc=$?; tail -1 $O/bench_dist1.log > $O/r02_bench_dist_rccl_world1_line.json
MBPLS_BENCH_SHARED_GPU=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --p 200000 --steps 2 --warmup 1 > $O/bench_shared2.log 2>&1; echo shared2 rc=$?; tail -1 $O/bench_shared2.log > $O/r02_bench_dist_shared_gpu_world2_line.json
for c in c3 c4 simpls nipals_q10 unipals_q10 nipals_mb; do python tools/bench_configs.py $c 2>/dev/null | tail -1 >> $O/r02_other_configs_lines.jsonl; done
MBPLS_KERNEL_IMPLICIT=1 python tools/bench_configs.py c4 2>/dev/null | tail -1 | sed 's/"C4 KERNEL Gram"/"C4 KERNEL Gram, never-writes-X (MBPLS_KERNEL_IMPLICIT=1)"/' >> $O/r02_other_configs_lines.jsonl
tools/micro/read_ceiling > $O/r02_read_ceiling.txt 2>&1
python tools/time_shapes.py 8 NIPALS > $O/r02_shapes.txt 2>&1; python tools/time_shapes.py 4 UNIPALS >> $O/r02_shapes.txt 2>&1; python -W ignore tools/time_sparse_shapes.py 2>&1 | grep '^n=' >> $O/r02_shapes.txt
python tools/time_long.py 1000000 2000 > $O/r02_long_columns.txt 2>&1; python tools/time_long.py 16384 100000 >> $O/r02_long_columns.txt 2>&1; python tools/time_small.py > $O/r02_small_problems.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 6 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/bench_under_rocprof.log 2>&1; echo stats rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/FETCH_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_fetch.log 2>&1; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/WRITE_SIZE -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-cols 0 --e2e-cols 0 --no-extras > $O/pmc_write.log 2>&1; echo write rc=$?
cd $R
find $O -name "*.csv" | head -20
ls -la $O
