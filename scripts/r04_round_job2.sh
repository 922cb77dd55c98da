# GPU suite + bench line of the build with gmg_config::block_fine (level 0 of kNN operators blocked)
O=$GRAFT_REPO_ROOT/gpurun_out/r04m; mkdir -p $O
timeout -s KILL 1500 python -m pytest tests -m gpu -q --tb=short --durations=8 2>&1 | tail -60 > $O/pytest_gpu_summary.txt
tail -25 $O/pytest_gpu_summary.txt
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err
tail -3 $O/bench.err; python - <<'PY'
import json, os
p = json.load(open(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r04m/bench.json"))
print("value", p["value"], "roofline", p["roofline"]["frac"], p["roofline"]["cycle"]["frac"])
v = p["variants"]["pointcloud_2M_knn8"]
print({k: v[k] for k in ("ms_per_step", "iterations_to_1e-4", "solve_ms", "second_solve_ms", "set_system_ms", "colors", "level0_sweep", "cycle_frac_of_peak", "level0_colour_major")})
PY
