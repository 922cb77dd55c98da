R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03e; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_setup.py tests/test_gpu_parity.py tests/test_dropin_api.py tests/test_gpu_hierarchy.py -m gpu -q -x 2>&1 | tail -5
show() { python -c "
import json,sys; j=json.loads(open('$1').read()); r=j['roofline']; print('$2', round(j['value'],4), j['config']['colors'], j['config']['levels'], j['iterations_to_1e-4'], 'set_system', round(j['set_system_ms'],1), 'sweep', round(r['launch_ms']*1e3,2), 'x', r['launches_per_sweep'], {k: round(v['ms']*1e3,1) for k,v in r['other_fine_kernels'].items()})"; }
for C in 4r 3; do
  python bench.py --config $C --cpu-cycles 0 --steps 30 2>/dev/null > $O/bench_$C.json; show $O/bench_$C.json "$C auto"
  for Q in 131072 65536; do GMG_QUAD_LEVEL_ROWS=$Q python bench.py --config $C --cpu-cycles 0 --steps 30 2>/dev/null > $O/bench_${C}_q$Q.json; show $O/bench_${C}_q$Q.json "$C quad<$Q"; done
done | tee $O/summary.txt
GMG_SETUP_TRACE=1 python bench.py --config 4r --cpu-cycles 0 --steps 5 2>&1 >/dev/null | grep "gmg setup" | tail -60 > $O/setup_trace_4r.txt
