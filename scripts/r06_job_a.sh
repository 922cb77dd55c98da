# round 6, first contact: new tests (RCCL on one rank, same-device refusal, fences), the device-built coarse inverse, bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06a; mkdir -p $O; cd $R
timeout -s KILL 900 python -m pytest tests/test_gpu_p2p.py -m gpu -q --tb=short -x -k "rccl or refused or (ipc_handles and poisson and 2)" 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -30 > $O/pytest_new.txt
tail -5 $O/pytest_new.txt
timeout -s KILL 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_setup.py tests/test_gpu_cycle_model.py tests/test_gpu_fullsize.py -m gpu -q --tb=short 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -40 > $O/pytest_core.txt
tail -8 $O/pytest_core.txt
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo; tail -5 $O/bench.err
python - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"]
try:
    d = json.loads(open(R + "/gpurun_out/r06a/bench.json").read().strip().splitlines()[-1])
    print(d["value"], d["set_system_ms"], d.get("set_system_cold_ms"), d["solver_timing_ms"]["solver_total"], d["roofline"]["frac"], d["roofline"]["cycle"]["frac"])
    print([ (l["level"], round(l["ms"],4)) for l in d["roofline"]["levels"]])
    print({k: (round(v["ms_per_step"], 4), v.get("set_system_ms")) for k, v in d["variants"].items() if "ms_per_step" in v})
except Exception as e:
    print("no bench line", e)
PY
