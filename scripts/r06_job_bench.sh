# bench line of the current build; argument: tag
R=$GRAFT_REPO_ROOT; T=${1:-e}; O=$R/gpurun_out/r06$T; mkdir -p $O; cd $R
timeout -s KILL 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
python - "$O/bench.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "set_system", d["set_system_ms"], "cold", d.get("set_system_cold_ms"), "solver_total", d["solver_timing_ms"]["solver_total"], "frac", d["roofline"]["frac"], "cycle frac", d["roofline"]["cycle"]["frac"], "iters", d["iterations_to_1e-4"])
print([(l["level"] if isinstance(l["level"], int) else l["level"][:12], round(l["ms"], 4)) for l in d["roofline"]["levels"]])
for k, v in d["variants"].items():
    if "ms_per_step" in v: print(k, round(v["ms_per_step"], 4), v.get("set_system_ms"), v.get("iterations_to_1e-4"))
    else: print(k, {kk: (round(vv["ms_per_step"], 4), vv.get("set_system_ms")) for kk, vv in v.items() if isinstance(vv, dict) and "ms_per_step" in vv})
PY
