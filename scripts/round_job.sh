# Evidence run of a round (one gpurun call): GPU suite, bench line, rocprofv3 stats / timelines / PMC for the bench workload and the
# other orderings / configs, size scaling, two ranks on one GPU, set-up and hierarchy breakdowns.  Usage: bash scripts/round_job.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03$TAG; mkdir -p $O; rm -f $R/gpurun_out/fullsize_configs.jsonl $R/gpurun_out/sor_default.jsonl
cd $R
timeout 1800 python -m pytest tests -m gpu -q --durations=6 > $O/pytest_gpu.log 2>&1; tail -12 $O/pytest_gpu.log
cp gpurun_out/fullsize_configs.jsonl $O/ 2>/dev/null; cp gpurun_out/sor_default.jsonl $O/ 2>/dev/null
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
python bench.py --n1 2829 --n2 2829 --cpu-cycles 0 --no-variants > $O/bench_8m.json 2> $O/bench_8m.err
python bench.py --n1 4483 --n2 4483 --cpu-cycles 0 --no-variants > $O/bench_20m.json 2> $O/bench_20m.err
cd /tmp; export TMPDIR=/tmp
prof() {   # name, bench args...
  N=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -- python $R/bench.py --steps 20 --warmup 3 --cpu-cycles 0 --no-variants "$@" > $O/bench_prof_$N.json 2> $O/prof_$N.log
  T=$(ls $O/prof_$N/*/*kernel_trace.csv | head -1); python $R/scripts/trace_cycle.py $T > $O/cycle_timeline_$N.txt 2>&1
  cp $(ls $O/prof_$N/*/*kernel_stats.csv | head -1) $O/kernel_stats_$N.csv; rm -rf $O/prof_$N
}
prof 3m; prof 3m_random --config 4r; prof pointcloud --config 3; prof 3m_smoothing_d3 --config 4s
pmc() {    # name, counters, bench args...
  N=$1; C=$2; shift; shift
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -- python $R/bench.py --steps 4 --warmup 1 --cpu-cycles 0 --no-variants --kernel-reps 4 "$@" > /dev/null 2>&1
  python $R/scripts/pmc_summary.py $(find $O/pmc_$N -name "*counter_collection.csv" | head -1) > $O/pmc_$N.txt; rm -rf $O/pmc_$N
}
pmc fetch_3m FETCH_SIZE; pmc write_3m WRITE_SIZE
pmc fetch_3m_random FETCH_SIZE --config 4r; pmc write_3m_random WRITE_SIZE --config 4r
pmc fetch_pointcloud FETCH_SIZE --config 3; pmc write_pointcloud WRITE_SIZE --config 3
pmc tcc_3m "TCC_HIT_sum TCC_MISS_sum"; pmc tcc_3m_random "TCC_HIT_sum TCC_MISS_sum" --config 4r; pmc tcc_pointcloud "TCC_HIT_sum TCC_MISS_sum" --config 3
cat $O/pmc_fetch_3m.txt $O/pmc_write_3m.txt > $O/pmc_fetch_write_summary.txt
cd $R
for S in 2 1; do GMG_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --shard-levels $S 2>$O/dist_shard$S.err | tail -1 > $O/bench_2ranks_1gpu_shard$S.json; done
python scripts/setup_breakdown.py 2>&1 | grep -A3 "^set_system" > $O/setup_breakdown_natural.txt
python scripts/setup_breakdown.py random 2>&1 | grep -A3 "^set_system" > $O/setup_breakdown_random.txt
python scripts/hierarchy_timing.py 2>&1 | tail -2 > $O/hierarchy_timing.txt; python scripts/hierarchy_timing.py random 2>&1 | tail -2 >> $O/hierarchy_timing.txt
python scripts/ldlt_team_bench.py scripts/micro/coarse_4.npz scripts/micro/coarse_4r.npz scripts/micro/coarse_3.npz 2>&1 | grep "ldlt\]" > $O/ldlt_team.txt
find $O -name "*.err" -size -1k -delete; du -sh $O; ls $O
