R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
for v in 0 1 0; do GMG_SETUP_TRACE=1 GMG_HIERARCHY_HOST_CLUSTER=$v python scripts/hierarchy_timing.py 2>&1 | grep -E "clustering on|^natural"; done | tee $O/hierarchy_timing.txt
GMG_SETUP_TRACE=1 python scripts/hierarchy_timing.py random 2>&1 | grep -E "clustering on|^random" | tee -a $O/hierarchy_timing.txt
