# 4 and 8 ranks on ONE GPU with the host side confined to 16 CPUs (the driver's container quota): functional N > 1 lines + what the spinning host threads cost there
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05s; mkdir -p $O
cd $R
nproc; taskset -c 0-15 nproc
for N in 4 8; do
GMG_DIST_BACKEND=gloo GMG_BENCH_NO_HALO_VARIANT=1 GMG_P2P_TIMEOUT_S=60 timeout -s KILL 1500 taskset -c 0-15 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2955$N bench.py --gpus $N --steps 5 --warmup 2 --cpu-cycles 0 2>$O/dist_$N.err | tail -1 > $O/bench_${N}ranks_1gpu_16cpus.json
python - $N <<'PY'
import json, os, sys
try:
    p = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r05s/bench_%sranks_1gpu_16cpus.json" % sys.argv[1]))
    print({k: p.get(k) for k in ("n_gpus", "value", "exchange", "iterations_to_1e-4", "host_threads_per_rank", "device_bytes_per_rank", "single_gpu_residues_reproduced")})
    print({k: (v.get("ms_per_step"), v.get("iterations_to_1e-4")) for k, v in p["variants"].items()}); print(p["setup"]["partitioned"], p["setup"]["whole_operator"]["set_system_ms"])
except Exception as e:
    print("no line", e)
PY
tail -3 $O/dist_$N.err
done
