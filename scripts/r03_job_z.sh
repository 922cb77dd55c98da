R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_hierarchy.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -2
( python scripts/hierarchy_timing.py 2>&1 | grep -E "^natural"; python scripts/hierarchy_timing.py 2>&1 | grep -E "^natural" ) | cut -c1-330 | tee $O/hierarchy_timing.txt
