R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -1
( python scripts/ab_cycle.py --config 4s --label "G6"; python scripts/ab_cycle.py --config 1 --label "G6" ) 2>/dev/null | cut -c1-200
