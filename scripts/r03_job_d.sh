R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03d; mkdir -p $O; cd $R
for C in 4r 3; do
python scripts/exp_base_order.py $C 2>&1 | tail -3
for ORD in none bfs grid rcm; do
  if [ $ORD = none ]; then unset GMG_EXP_BASE_ORDER_FILE; else export GMG_EXP_BASE_ORDER_FILE=/tmp/order_$ORD.bin; fi
  python bench.py --config $C --cpu-cycles 0 --steps 30 2>/dev/null > $O/bench_${C}_$ORD.json
  python -c "
import json; j=json.loads(open('$O/bench_${C}_$ORD.json').read()); r=j['roofline']; print('$C $ORD', round(j['value'],4), j['config']['colors'], j['iterations_to_1e-4'], round(j['set_system_ms'],1), 'sweep launch avg', round(r['launch_ms']*1e3,2), 'x', r['launches_per_sweep'], {k: round(v['ms']*1e3,1) for k,v in r['other_fine_kernels'].items()})"
done; done | tee $O/summary.txt
