R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03g; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 20 --warmup 3 --cpu-cycles 0 --no-variants > $O/bench.json 2> $O/prof.log
T=$(ls $O/prof/*/*kernel_trace.csv | head -1); python $R/scripts/trace_cycle.py $T > $O/timeline.txt 2>&1; cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; rm -rf $O/prof
cat $O/timeline.txt
