R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05u; mkdir -p $O; cd $R
timeout -s KILL 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
for N in 2829 4483; do
timeout -s KILL 900 python bench.py --n1 $N --n2 $N --no-variants --cpu-cycles 0 > $O/bench_$N.json 2> $O/bench_$N.err
python - $N <<'PY'
import json, os, sys
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05u/bench_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print(d["config"]["n_vertices"], d["value"], d["iterations_to_1e-4"], d["set_system_ms"], d["set_system_cold_ms"], d["structure_prepare_ms"], d["solver_timing_ms"]["solver_total"], d["roofline"]["frac"], d["roofline"]["cycle"]["frac"])
PY
done
