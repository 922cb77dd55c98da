R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05n; mkdir -p $O
cd $R
timeout -s KILL 1800 python -m pytest tests/test_gpu_p2p.py -m gpu -q --tb=short -x -k "ipc_handles or collective or missing_peer or multigridsolver" 2>&1 | tail -5
for i in 1 2; do
GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>$O/dist.err | tail -1 > $O/bench_2ranks_1gpu_$i.json
python - $i <<'PY'
import json, os, sys
p = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05n/bench_2ranks_1gpu_%s.json" % sys.argv[1]))
print({k: p.get(k) for k in ("value", "exchange", "iterations_to_1e-4")}); print({k: (v.get("ms_per_step"), v.get("samples_ms"), v.get("default_again"), v.get("exchange_launches_per_cycle")) for k, v in p["variants"].items()})
PY
done
