# round 5, job q: the evidence run of the round -- full GPU suite, bench line with variants, rocprofv3 stats / timelines for the bench workload and the other
# configs, set-up timelines, two ranks on one GPU, FETCH / WRITE counters of the bench workload's kernels
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05q; mkdir -p $O
cd $R
timeout -s KILL 2700 python -m pytest tests -m gpu -q --tb=short --durations=10 -s 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -60 > $O/pytest_gpu_summary.txt
grep -n "passed\|failed\|^FAILED" $O/pytest_gpu_summary.txt | head -20
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
bash scripts/r05_prof.sh q 3m 3m_random:--config:4r pointcloud:--config:3 3m_smoothing_d3:--config:4s 3m_bilaplacian:--config:5b 722k:--config:2 > /dev/null 2>&1
cd $R
GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>$O/dist.err | tail -1 > $O/bench_2ranks_1gpu.json
cd /tmp; export TMPDIR=/tmp
for ord in natural random; do
timeout -s KILL 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_$ord -- python $R/scripts/setup_trace.py $ord > $O/setup_trace_$ord.txt 2> $O/prof_$ord.log
K=$(ls $O/prof_$ord/*/*kernel_trace.csv | head -1); M=$(ls $O/prof_$ord/*/*memory_copy_trace.csv | head -1)
python $R/scripts/setup_timeline.py $K $M > $O/setup_timeline_$ord.txt 2>&1
rm -rf $O/prof_$ord
done
cd $R
for ord in natural random; do GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py $ord 2>&1 | grep SETUP_TRACE > $O/setup_trace_${ord}_unprofiled.txt; done
bash scripts/r05_pmc.sh q > $O/pmc_lines.txt 2>&1
find $O -name "*.err" -size -1k -delete; find $O -name "*.log" -delete; ls $O
