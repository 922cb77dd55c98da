# round 5, job h: full GPU suite, bench line with variants, rocprofv3 stats / timelines for the bench workload and the other configs, two ranks on one GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05h; mkdir -p $O
cd $R
timeout -s KILL 2400 python -m pytest tests -m gpu -q --tb=short --durations=10 -s 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -60 > $O/pytest_gpu_summary.txt
grep -n "merged classes\|passed\|failed\|^FAILED" $O/pytest_gpu_summary.txt | head -20
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo; grep "variant" $O/bench.err
bash scripts/r05_prof.sh h 3m 3m_random:--config:4r pointcloud:--config:3 3m_smoothing_d3:--config:4s 3m_bilaplacian:--config:5b 722k:--config:2 > /dev/null 2>&1
cd $R
for S in 2; do GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --shard-levels $S 2>$O/dist_shard$S.err | tail -1 > $O/bench_2ranks_1gpu_shard$S.json; done
GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py natural 2>&1 | grep SETUP_TRACE > $O/setup_trace_natural.txt
GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py random 2>&1 | grep SETUP_TRACE > $O/setup_trace_random.txt
find $O -name "*.err" -size -1k -delete; find $O -name "*.log" -delete; ls $O
