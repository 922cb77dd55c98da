R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03w; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
( python scripts/ab_cycle.py --config 4r --label "4r"; python scripts/ab_cycle.py --config 3 --label "3"; python scripts/ab_cycle.py --config 4 --label "4" ) 2>/dev/null | tee $O/ab.jsonl
