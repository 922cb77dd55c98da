R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03o; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cycle_model.py tests/test_gpu_p2p.py tests/test_gpu_mixed.py tests/test_gpu_stress.py -m gpu -q -x 2>&1 | tail -3
( python scripts/ab_cycle.py --config 4 --label "publish folded"
  GMG_NO_PUBLISH_FOLD=1 python scripts/ab_cycle.py --config 4 --label "separate publish kernel"
  python scripts/ab_cycle.py --config 4 --label "publish folded again"
  python scripts/ab_cycle.py --config 4s --label "d=3 with team columns" ) 2>/dev/null | tee $O/ab.jsonl
