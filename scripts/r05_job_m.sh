R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05m; mkdir -p $O
cd $R
timeout -s KILL 1500 python -m pytest tests/test_gpu_setup.py tests/test_gpu_parity.py tests/test_gpu_mixed.py tests/test_gpu_hierarchy.py tests/test_dropin_api.py -m gpu -q --tb=short -x 2>&1 | tail -12
for i in 1 2; do GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py natural 2>&1 | grep SETUP_TRACE; done
GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py random 2>&1 | grep SETUP_TRACE
timeout -s KILL 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
