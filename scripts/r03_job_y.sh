R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03y; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_setup.py -m gpu -q -x -k "16_bit" 2>&1 | tail -12
python - <<PY 2>&1 | grep -v amdgpu.ids | tee $O/col16_from.txt
import sys, os, time; sys.path.insert(0, '.')
import numpy as np
import bench
from gravo_mg_amd import cabi
def run(tag, H, mass, lhs, rhs):
    ref = None
    for no16 in (0, 1):
        if no16: os.environ["GMG_NO_COL16"] = "1"
        eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
        os.environ.pop("GMG_NO_COL16", None)
        keys = ("col16_l0", "col16_R_l0", "col16_P_l0")
        out = {"case": tag, "A/R/P": [eng.timing(k) for k in keys]}
        if not no16: out["failed"] = [eng.timing(k + "_failed_slices") for k in keys]; out["windows"] = [eng.timing(k + "_windows") for k in keys]; out["mode"] = [eng.timing(k + "_mode") if eng.timing(k) else 0 for k in keys]; out["slices"] = eng.level_info(0)["n_pad"] // 64
        d = rhs.shape[1]
        for name, kind in (("sweep", 0), ("residual", 1), ("restrict", 2), ("prolong", 3), ("norm", 4)):
            ms, launches = eng.bench_kernel(kind, 0, d, 100)
            out[name] = round(1e3 * ms / (launches if kind == 0 else 1), 2)
        eng.load_problem(rhs, rhs); eng.run_cycles(5, 2)
        t = time.perf_counter(); eng.run_cycles(50, 2); out["cycle_ms"] = round(20 * (time.perf_counter() - t), 4)
        eng.load_problem(rhs, rhs); hist = eng.run_cycles(3, 2); x = eng.fetch_solution()
        if no16 == 0: ref = (hist, x)
        else: out["bitwise_equal"] = bool(np.array_equal(hist, ref[0]) and np.array_equal(x, ref[1]))
        print(out, flush=True)
run("3M", *bench.build_workload(1732, 1732, "natural"))
run("8M", *bench.build_workload(2829, 2829, "natural"))
for cfg in ("4r", "3", "4s", "5"):
    name, H, mass, lhs, rhs = bench.build_config(cfg)
    run(cfg, H, mass, lhs, rhs)
PY
