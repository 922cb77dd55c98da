R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R; rm -f gpurun_out/sor_default.jsonl gpurun_out/fullsize_configs.jsonl
timeout 2400 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -25
cat gpurun_out/sor_default.jsonl
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json; echo
