R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
python - <<PY 2>&1 | grep -v amdgpu.ids | tee $O/col16.txt
import sys, os, time; sys.path.insert(0, '.')
import numpy as np
import bench
from gravo_mg_amd import cabi
H, mass, lhs, rhs = bench.build_workload(1732, 1732, "natural")
ref = {}
for no16 in (0, 1, 0):
    if no16: os.environ["GMG_NO_COL16"] = "1"
    eng = cabi.Engine(); eng.use_hierarchy(H); eng.set_mass(mass); eng.set_system(lhs)
    os.environ.pop("GMG_NO_COL16", None)
    out = {"col16": eng.timing("col16_l0"), "A/R/P": [eng.timing(k) for k in ("col16_l0", "col16_R_l0", "col16_P_l0")], "failed": [eng.timing(k + "_failed_slices") for k in ("col16_l0", "col16_R_l0", "col16_P_l0")] if not no16 else None, "set_system": round(eng.timing("setup_total"), 2)}
    for name, kind in (("sweep", 0), ("residual", 1), ("restrict", 2), ("prolong", 3), ("norm", 4)):
        ms, launches = eng.bench_kernel(kind, 0, 1, 200)
        out[name] = (round(1e3 * ms / (launches if kind == 0 else 1), 2), launches)
    eng.load_problem(rhs, rhs); eng.run_cycles(5, 2)
    t = time.perf_counter(); eng.run_cycles(100, 2); out["cycle_ms"] = round(10 * (time.perf_counter() - t), 4)
    eng.load_problem(rhs, rhs); hist = eng.run_cycles(4, 2); x = eng.fetch_solution()
    if no16 == 0: ref = (hist, x)
    else: out["bitwise_equal"] = bool(np.array_equal(hist, ref[0]) and np.array_equal(x, ref[1]))
    print(out, flush=True)
PY
timeout 1500 python -m pytest tests/test_gpu_setup.py tests/test_gpu_parity.py tests/test_gpu_cycle_model.py tests/test_gpu_mixed.py tests/test_gpu_p2p.py tests/test_gpu_dist.py -m gpu -q -x 2>&1 | tail -4
( for c in 4 4r 3 5; do python scripts/ab_cycle.py --config $c --label "col16"; GMG_NO_COL16=1 python scripts/ab_cycle.py --config $c --label "int32"; done ) 2>/dev/null | tee $O/ab.jsonl
