R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02p; mkdir -p $O; rm -f $R/gpurun_out/fullsize_configs.jsonl
cd $R
timeout 1500 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 400 $O/bench.json; echo
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3m -- python $R/bench.py --steps 20 --warmup 3 --cpu-cycles 0 --no-variants > $O/prof3m.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 4 --warmup 1 --cpu-cycles 0 --no-variants --kernel-reps 4 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 4 --warmup 1 --cpu-cycles 0 --no-variants --kernel-reps 4 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3m_random -- python $R/bench.py --steps 20 --warmup 3 --cpu-cycles 0 --no-variants --order random > $O/prof3m_random.log 2>&1
python $R/bench.py --n1 2829 --n2 2829 --cpu-cycles 0 --no-variants > $O/bench_8m.json 2> $O/bench_8m.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof8m -- python $R/bench.py --n1 2829 --n2 2829 --steps 20 --warmup 3 --cpu-cycles 0 --no-variants > $O/prof8m.log 2>&1
python $R/bench.py --n1 4483 --n2 4483 --cpu-cycles 0 --no-variants > $O/bench_20m.json 2> $O/bench_20m.err
for S in 2 1; do GMG_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 2 --steps 10 --warmup 2 --shard-levels $S 2>$O/dist_shard$S.err | tail -1 > $O/dist_shard$S.json; done
python $R/scripts/coarse_host_time.py 2>&1 | tail -1 > $O/coarse_host_time.txt
python $R/scripts/hierarchy_timing.py 2>&1 | tail -2 > $O/hierarchy_timing.txt
python $R/scripts/l1_sweep_ab.py 2>&1 | tail -2 > $O/l1_sweep_ab.txt
ls $O
