R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
( for c in 4 4r 3 4s 5; do python scripts/ab_cycle.py --config $c --label "codes"; done ) 2>/dev/null | cut -c1-330 | tee $O/ab_codes.jsonl
