# last build of the round: full GPU suite + bench line + two ranks on one GPU
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05x; mkdir -p $O; cd $R
timeout -s KILL 2700 python -m pytest tests -m gpu -q --tb=short --durations=8 -s 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -40 > $O/pytest_gpu_summary.txt
grep -n "passed\|failed\|^FAILED" $O/pytest_gpu_summary.txt | head
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; echo
GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>$O/dist.err | tail -1 > $O/bench_2ranks_1gpu.json
python - <<'PY'
import json, os
R = os.environ["GRAFT_REPO_ROOT"]
d = json.loads(open(R + "/gpurun_out/r05x/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["set_system_ms"], d["set_system_cold_ms"], d["solver_timing_ms"]["solver_total"], d["roofline"]["frac"], d["roofline"]["cycle"]["frac"])
print({k: (round(v["ms_per_step"], 4) if "ms_per_step" in v else None) for k, v in d["variants"].items()})
p = json.load(open(R + "/gpurun_out/r05x/bench_2ranks_1gpu.json"))
print(p["value"], p["exchange"], {k: v.get("ms_per_step") for k, v in p["variants"].items()}, p["setup"]["partitioned"]["set_system_ms"])
PY
