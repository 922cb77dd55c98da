# round 5, job c: GPU suite + set-up trace + bench of the build with the structure prepared at hierarchy time and the env/test-hook clean-up
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05c; mkdir -p $O
cd $R
timeout -s KILL 1800 python -m pytest tests -m gpu -q --tb=short --durations=8 -x 2>&1 | tail -60 > $O/pytest_gpu_summary.txt
tail -12 $O/pytest_gpu_summary.txt
GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py natural > $O/setup_trace_natural.txt 2>&1; grep SETUP_TRACE $O/setup_trace_natural.txt
timeout -s KILL 900 python bench.py --no-variants > $O/bench_novar.json 2> $O/bench_novar.err; tail -c 200 $O/bench_novar.json; echo
python - <<'PY'
import json, os
p = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05c/bench_novar.json"))
print({k: p[k] for k in ("value", "set_system_ms", "set_system_structure_prepared", "structure_prepare_ms", "use_hierarchy_ms", "set_system_cold_ms", "solve_ms", "iterations_to_1e-4")})
print(p["solver_timing_ms"])
PY
