# round 3, job b: GPU suite after the advice fixes; kernel timelines + PMC of the random-order mesh and the point cloud
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 > $O/pytest_gpu.log 2>&1; tail -8 $O/pytest_gpu.log
cd /tmp; export TMPDIR=/tmp
for C in 4r 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$C -- python $R/bench.py --config $C --steps 20 --warmup 3 --cpu-cycles 0 > $O/bench_$C.json 2> $O/prof_$C.log
  T=$(ls $O/prof_$C/*/*kernel_trace.csv | head -1); python $R/scripts/trace_cycle.py $T > $O/timeline_$C.txt 2>&1
  cp $(ls $O/prof_$C/*/*kernel_stats.csv | head -1) $O/kernel_stats_$C.csv
done
for C in 4r 4; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_$C -- python $R/bench.py --config $C --steps 4 --warmup 1 --cpu-cycles 0 --kernel-reps 4 > /dev/null 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/pmc_tcc_$C -- python $R/bench.py --config $C --steps 4 --warmup 1 --cpu-cycles 0 --kernel-reps 4 > /dev/null 2>&1
done
find $O -name "*counter_collection.csv" | head; du -sh $O
# keep the merged output small: drop the raw traces, keep counter csvs
find $O -name "*kernel_trace.csv" -path "*prof_*" -delete
for f in $(find $O -name "*counter_collection.csv"); do python $R/scripts/pmc_summary.py $f > $O/$(basename $(dirname $(dirname $f)))_summary.txt; done
find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; du -sh $O
