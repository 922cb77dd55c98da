# rocprofv3 kernel trace + stats of bench.py for named configs: bash scripts/r06_prof.sh <tag> [3m] [3m_smoothing_d3:--config:4s] ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  N=${spec%%:*}; A=""; [ "$spec" != "$N" ] && A=$(echo ${spec#*:} | tr ':' ' ')
  timeout -s KILL 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$N -- python $R/bench.py --steps 20 --warmup 3 --cpu-cycles 0 --no-variants $A > $O/bench_prof_$N.json 2> $O/prof_$N.log
  T=$(ls $O/prof_$N/*/*kernel_trace.csv | head -1); python $R/scripts/trace_cycle.py $T > $O/cycle_timeline_$N.txt 2>&1
  cp $(ls $O/prof_$N/*/*kernel_stats.csv | head -1) $O/kernel_stats_$N.csv; rm -rf $O/prof_$N
  tail -c 300 $O/bench_prof_$N.json; echo
done
