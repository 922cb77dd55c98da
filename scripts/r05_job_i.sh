R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05i; mkdir -p $O
cd $R
timeout -s KILL 900 python scripts/tiny_colours_ab.py 2>/dev/null | grep workload > $O/tiny_colours_ab.jsonl; cat $O/tiny_colours_ab.jsonl
timeout -s KILL 600 python -m pytest tests/test_gpu_setup.py -m gpu -q --tb=short -k "tiny" -s 2>&1 | grep "merged classes\|passed\|failed" | head
GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 2>$O/dist.err | tail -1 > $O/bench_2ranks_1gpu.json
python - <<'PY'
import json, os
p = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05i/bench_2ranks_1gpu.json"))
print({k: p.get(k) for k in ("value", "exchange", "device_bytes_per_rank", "iterations_to_1e-4", "single_gpu_residues_reproduced")})
print(p.get("setup")); print(p.get("cpu_baseline", {}).get("value")); print(p.get("variants"))
PY
tail -3 $O/dist.err
