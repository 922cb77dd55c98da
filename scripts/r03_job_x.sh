R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03x; mkdir -p $O; cd $R
GMG_SETUP_TRACE=1 python scripts/setup_breakdown.py random 2>&1 | grep -E "gmg setup|^set_system|timeline" | head -80 | cut -c1-400 | tee $O/trace_random.txt
