R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03c; mkdir -p $O; cd $R
for TP in 6000 1000000000; do echo "== GMG_LDLT_TEAM_PANEL=$TP"; GMG_LDLT_TEAM_PANEL=$TP python scripts/ldlt_team_bench.py scripts/micro/coarse_4.npz scripts/micro/coarse_4r.npz scripts/micro/coarse_3.npz 2>&1 | grep ldlt; done | tee $O/ldlt_team.txt
for P in 2 4 8; do echo "== GMG_LDLT_PARTS=$P"; GMG_LDLT_PARTS=$P python scripts/ldlt_team_bench.py scripts/micro/coarse_4.npz scripts/micro/coarse_4r.npz 2>&1 | grep "threads:\|parts of"; done | tee $O/ldlt_parts.txt
python bench.py --cpu-cycles 0 --steps 50 2>/dev/null > $O/bench.json; python -c "
import json; j=json.loads(open('$O/bench.json').read()); print(j['value'], j['solver_timing_ms']['coarse_host_ms'], {k:(v['ms_per_step']) for k,v in j['variants'].items()})"
