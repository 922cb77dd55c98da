R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
ulimit -c 0
for i in 1 2 3 4 5 6; do GMG_SEGV_BACKTRACE=1 timeout 600 python scripts/r03_repro.py > $O/rp$i.out 2> $O/rp$i.err; echo "run $i rc=$? $(tail -1 $O/rp$i.out)"; done
for i in 1 2 3 4; do timeout 600 python scripts/parity_sweep.py > $O/ps$i.out 2> $O/ps$i.err; echo "sweep $i rc=$? $(tail -1 $O/ps$i.out | cut -c1-60)"; done
