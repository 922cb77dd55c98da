# round 5, job d: GPU suite (partitioned set-up tests incl. the 3 M memory bound) + two ranks on one GPU through bench.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05d; mkdir -p $O
cd $R
timeout -s KILL 2400 python -m pytest tests -m gpu -q --tb=short --durations=12 -x 2>&1 | tail -70 > $O/pytest_gpu_summary.txt
tail -25 $O/pytest_gpu_summary.txt
cp gpurun_out/partitioned_setup_memory.json $O/ 2>/dev/null
for S in 2; do GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --shard-levels $S 2>$O/dist_shard$S.err | tail -1 > $O/bench_2ranks_1gpu_shard$S.json; done
tail -5 $O/dist_shard2.err
python - <<'PY'
import json, os
p = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05d/bench_2ranks_1gpu_shard2.json"))
print({k: p.get(k) for k in ("value", "exchange", "exchange_note", "device_bytes_per_rank", "device_bytes_per_rank_peak_during_setup", "iterations_to_1e-4", "single_gpu_residues_reproduced")})
print(p.get("setup")); print(p.get("cpu_baseline", {}).get("value")); print({k: v.get("ms_per_step") for k, v in p.get("variants", {}).items()})
PY
