R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py --cpu-cycles 0 2> $O/bench.err > $O/bench.json; python - <<PY
import json; j=json.load(open("$O/bench.json")); r=j["roofline"]
print(j["value"], j["iterations_to_1e-4"], {k: r[k] for k in ("kernel","frac","launch_ms","stored_format_frac","index_format")}, {k:(round(v["ms"]*1e3,1), round(v["GBps"])) for k,v in r["other_fine_kernels"].items()}, r["cycle"]["frac"])
print({k: (v.get("ms_per_step") or v.get("value")) for k, v in j["variants"].items()})
print(j["solver_timing_ms"], j["set_system_ms"])
PY
