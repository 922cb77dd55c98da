R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05r; mkdir -p $O
cd $R
timeout -s KILL 600 python scripts/colour_class_sizes.py 2>&1 | grep -v "^\[" | tail -5
timeout -s KILL 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cycle_model.py tests/test_gpu_setup.py tests/test_gpu_sor_default.py -m gpu -q --tb=short -x 2>&1 | tail -4
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err
python - <<'PY'
import json, os
d = json.loads(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05r/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["set_system_ms"], d["solver_timing_ms"]["solver_total"])
for k, v in d["variants"].items():
    if "ms_per_step" in v: print(k, round(v["ms_per_step"], 4), v.get("iterations_to_1e-4"), v.get("colors"), round(v.get("set_system_ms", 0), 2), v.get("device_coarse_inverse"))
    else: print(k, {a: (round(b["ms_per_step"], 4), b.get("reach"), b.get("colors")) for a, b in v.items() if isinstance(b, dict)})
PY
