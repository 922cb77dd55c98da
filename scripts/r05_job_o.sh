R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05o; mkdir -p $O
cd $R
timeout -s KILL 600 python scripts/d3_cycle.py 2>&1 | tail -3
timeout -s KILL 600 python scripts/d3_cycle.py il_probe=1 2>&1 | tail -3
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cycle_model.py -m gpu -q --tb=short -x 2>&1 | tail -3
timeout -s KILL 600 python bench.py --no-variants --cpu-cycles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], [ (l['level'], round(l['ms'],4)) for l in d['roofline']['levels']])"
