R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_hierarchy.py -m gpu -q -x 2>&1 | tail -2
( python scripts/hierarchy_timing.py 2>&1 | grep -E "^natural"; GMG_HIERARCHY_FULL_DIJKSTRA=1 python scripts/hierarchy_timing.py 2>&1 | grep -E "^natural" | sed 's/^/full-dijkstra /'; python scripts/hierarchy_timing.py random 2>&1 | grep -E "^random" ) | tee $O/hierarchy_timing.txt
