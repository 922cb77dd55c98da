R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cycle_model.py tests/test_gpu_mixed.py -m gpu -q -x 2>&1 | tail -5
( python scripts/ab_cycle.py --config 4 --label "default"
  GMG_NO_DELTA_RESIDUAL=1 python scripts/ab_cycle.py --config 4 --label "no delta residual"
  python scripts/ab_cycle.py --config 4 --label "default again"
  GMG_QUAD_LEVEL_ROWS=131072 python scripts/ab_cycle.py --config 3 --label "quad<131072"
  python scripts/ab_cycle.py --config 3 --label "quad<262144 (default)"
  GMG_QUAD_LEVEL_ROWS=131072 python scripts/ab_cycle.py --config 3 --label "quad<131072 again"
  GMG_QUAD_LEVEL_ROWS=40000 python scripts/ab_cycle.py --config 4 --label "cfg4 quad<40000 (level 2 with 80k rows on ep)"
  GMG_LDLT_THREADS=2 python scripts/ab_cycle.py --config 4 --label "cfg4 ldlt threads 2"
  GMG_LDLT_THREADS=3 python scripts/ab_cycle.py --config 4 --label "cfg4 ldlt threads 3"
  GMG_LDLT_THREADS=2 python scripts/ab_cycle.py --config 4r --label "cfg4r ldlt threads 2"
  python scripts/ab_cycle.py --config 4r --label "cfg4r default"
) 2>/dev/null | tee $O/ab.jsonl
