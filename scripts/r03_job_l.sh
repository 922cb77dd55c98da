R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
python scripts/setup_breakdown.py > $O/setup_natural.txt 2>&1; tail -40 $O/setup_natural.txt
