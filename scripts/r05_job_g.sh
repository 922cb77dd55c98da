# round 5, job g: p2p + set-up tests after the collective parity fix and the merged tiny-colour launch; per-rank set-up cost of rank 0 of P
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05g; mkdir -p $O
cd $R
timeout -s KILL 1800 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_setup.py tests/test_gpu_multi_device.py -m gpu -q --tb=short --durations=6 -x -s 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -50 > $O/pytest_gpu_summary.txt
grep -n "merged classes\|passed\|failed\|Error" $O/pytest_gpu_summary.txt | head -20
for P in 2 4 8; do timeout -s KILL 600 python scripts/partition_probe.py $P 2>&1 | grep -v "^\[bench\]\|amdgpu.ids" | grep "whole\|rank0\|marks\|dist_plan" > $O/partition_probe_P$P.txt; cat $O/partition_probe_P$P.txt; done
