R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
timeout 1500 python -m pytest tests/test_dropin_api.py tests/test_gpu_parity.py tests/test_gpu_p2p.py -m gpu -q -x 2>&1 | tail -2
python bench.py --cpu-cycles 0 --no-variants 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print('solve_ms', j['solve_ms'], j['solver_timing_ms'], j['value'])"
python scripts/dropin_timing.py 2>&1 | tail -6
