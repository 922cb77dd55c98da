R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03j; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -12
