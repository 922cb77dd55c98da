# HBM traffic of a workload's kernels (rocprofv3 --pmc, ONE counter per pass, kernel trace only): bash scripts/r06_pmc.sh <tag> <name> [bench args]
TAG=$1; NAME=$2; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06$TAG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
pmc() { N=$1; C=$2; shift; shift
  timeout -s KILL 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_$N -- python $R/bench.py --steps 4 --warmup 1 --cpu-cycles 0 --no-variants --kernel-reps 4 "$@" > /dev/null 2> $O/pmc_$N.log
  python $R/scripts/pmc_summary.py $(find $O/pmc_$N -name "*counter_collection.csv" | head -1) > $O/pmc_$N.txt 2>&1; rm -rf $O/pmc_$N; }
pmc fetch_$NAME FETCH_SIZE "$@"; pmc write_$NAME WRITE_SIZE "$@"
cat $O/pmc_fetch_$NAME.txt $O/pmc_write_$NAME.txt > $O/pmc_fetch_write_$NAME.txt; rm -f $O/pmc_fetch_$NAME.txt $O/pmc_write_$NAME.txt
grep "gs_color<double, [13], 2\|gs_block_ep<double, [13]\|restrict_sweep0\|dense_symv" $O/pmc_fetch_write_$NAME.txt | cut -c1-130
