# soak_factor.py with a watchdog (Python stacks of all threads after 25 s without finishing a round)
O=$GRAFT_REPO_ROOT/gpurun_out/r04s; mkdir -p $O
GMG_SOAK_WATCHDOG=25 timeout -s KILL 200 python -X faulthandler scripts/soak_factor.py ${1:-300} > $O/soak_fixed.log 2>&1
echo "rc=$? rounds=$(grep -c 'concurrent round' $O/soak_fixed.log)"; grep -A12 "^Thread\|most recent call first" $O/soak_fixed.log | grep "File\|Thread" | cut -c1-160 | head -30; tail -2 $O/soak_fixed.log
