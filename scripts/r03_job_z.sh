R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
for i in 1 2; do python bench.py --n1 2829 --n2 2829 --cpu-cycles 0 --no-variants 2>$O/err$i.txt | python -c "
import sys,json; j=json.loads(sys.stdin.read()); print(j['set_system_ms'], j['solver_timing_ms'])"; grep "set_system" $O/err$i.txt | cut -c1-300; done
python scripts/setup_breakdown.py natural 2829 2>&1 | grep -A1 "^set_system" | head -2 | cut -c1-700
