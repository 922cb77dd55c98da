R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05v; mkdir -p $O; cd $R
timeout -s KILL 1800 python -m pytest tests/test_gpu_setup.py tests/test_gpu_parity.py tests/test_gpu_fine_blocks.py tests/test_gpu_hierarchy.py tests/test_dropin_api.py -m gpu -q --tb=short -x 2>&1 | tail -6
timeout -s KILL 1800 python -m pytest tests/test_gpu_p2p.py -m gpu -q --tb=short -x -k "ipc_handles or partitioned" 2>&1 | tail -4
timeout -s KILL 600 python bench.py --no-variants --cpu-cycles 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['set_system_ms'], d['set_system_cold_ms'], d['structure_prepare_ms'], d['use_hierarchy_ms'])"
timeout -s KILL 600 python bench.py --no-variants --cpu-cycles 0 --config 5b 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['set_system_ms'], d.get('set_system_cold_ms'))"
python scripts/partition_probe.py 2 2>&1 | tail -4 | cut -c1-700
