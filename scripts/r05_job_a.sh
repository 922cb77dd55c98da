# round 5, job a: baseline of the round-4 build -- GPU suite, bench line, set-up timelines (rocprofv3 kernel + memory-copy trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05a; mkdir -p $O
cd $R
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short --durations=8 -x 2>&1 | tail -40 > $O/pytest_gpu_summary.txt
tail -5 $O/pytest_gpu_summary.txt
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
for ORD in natural random; do
  GMG_SETUP_TRACE=1 timeout -s KILL 300 python scripts/setup_trace.py $ORD > $O/setup_trace_$ORD.txt 2>&1
  (cd /tmp; export TMPDIR=/tmp; timeout -s KILL 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_$ORD -- python $R/scripts/setup_trace.py $ORD > $O/prof_$ORD.log 2>&1)
  K=$(ls $O/prof_$ORD/*/*kernel_trace.csv | head -1); M=$(ls $O/prof_$ORD/*/*memory_copy_trace.csv | head -1)
  python scripts/setup_timeline.py $K $M > $O/setup_timeline_$ORD.txt 2>&1
  grep SETUP_TRACE $O/prof_$ORD.log >> $O/setup_timeline_$ORD.txt
  rm -rf $O/prof_$ORD
done
timeout -s KILL 200 python scripts/setup_breakdown.py 2>&1 | grep -A3 "^set_system" > $O/setup_breakdown_natural.txt
ls -la $O
