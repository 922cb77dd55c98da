R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cycle_model.py -m gpu -q -x 2>&1 | tail -3
( python scripts/ab_cycle.py --config 4s --label "d=3 smoothing 3M"
  python scripts/ab_cycle.py --config 4 --label "d=1 poisson 3M" ) 2>/dev/null | tee $O/ab.jsonl
python bench.py --config 4s --cpu-cycles 0 --steps 20 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('4s', round(j['value'],4), 'sweep launch', round(r['launch_ms']*1e3,2), {k: round(v['ms']*1e3,1) for k,v in r['other_fine_kernels'].items()})"
