R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05l; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for ord in natural random; do
timeout -s KILL 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_$ord -- python $R/scripts/setup_trace.py $ord > $O/setup_trace_$ord.txt 2> $O/prof_$ord.log
K=$(ls $O/prof_$ord/*/*kernel_trace.csv | head -1); M=$(ls $O/prof_$ord/*/*memory_copy_trace.csv | head -1)
python $R/scripts/setup_timeline.py $K $M > $O/setup_timeline_$ord.txt 2>&1
rm -rf $O/prof_$ord
grep SETUP_TRACE $O/setup_trace_$ord.txt | head -3; cat $O/setup_timeline_$ord.txt | cut -c1-200
done
