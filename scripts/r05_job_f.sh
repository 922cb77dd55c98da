# round 5, job f: collective-exchange tests (with details), partition memory bound, set-up trace with the speculative upload
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05f; mkdir -p $O
cd $R
timeout -s KILL 900 python -m pytest tests/test_gpu_p2p.py -m gpu -q --tb=short -k "collective or partitioned_set_up" 2>&1 | tail -60 > $O/pytest_coll.txt
grep -n "why_h\|AssertionError\|passed\|failed\|partitioned set-up" $O/pytest_coll.txt | head -30
cp gpurun_out/partitioned_setup_memory.json $O/ 2>/dev/null
GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py natural > $O/setup_trace_natural.txt 2>&1; grep SETUP_TRACE $O/setup_trace_natural.txt
timeout -s KILL 600 python scripts/partition_probe.py 2 > $O/partition_probe.txt 2>&1; grep -v "^\[bench\]\|amdgpu.ids" $O/partition_probe.txt | grep "whole\|rank0" 
