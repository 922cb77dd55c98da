R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03i; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_p2p.py -m gpu -q -x 2>&1 | tail -3
( python scripts/ab_cycle.py --config 4 --label "default (no block table)"
  GMG_EP_BLOCK_TABLE=1 python scripts/ab_cycle.py --config 4 --label "block table"
  python scripts/ab_cycle.py --config 4 --label "default again"
  python scripts/ab_cycle.py --config 3 --label "cfg3 default"
) 2>/dev/null | tee $O/ab.jsonl
