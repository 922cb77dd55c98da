R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -1
( python scripts/ab_cycle.py --config 4s --label "restrict G8"; python scripts/ab_cycle.py --config 4 --label "restrict G8" ) 2>/dev/null | cut -c1-200
