R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05k; mkdir -p $O
cd $R
timeout -s KILL 1500 python -m pytest tests/test_gpu_setup.py tests/test_gpu_parity.py tests/test_gpu_mixed.py -m gpu -q --tb=short -x 2>&1 | tail -6
for i in 1 2; do GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py natural 2>&1 | grep SETUP_TRACE; done
GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py random 2>&1 | grep SETUP_TRACE
GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>$O/dist.err | tail -1 > $O/bench_2ranks_1gpu.json
python - <<'PY'
import json, os
p = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05k/bench_2ranks_1gpu.json"))
print({k: p.get(k) for k in ("value", "exchange", "device_bytes_per_rank", "iterations_to_1e-4")}); print({k: v.get("ms_per_step") for k, v in p["variants"].items()})
PY
