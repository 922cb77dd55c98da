# round 5, job e: what a partitioned handle holds + the p2p / collective tests
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05e; mkdir -p $O
cd $R
timeout -s KILL 600 python scripts/partition_probe.py 2 > $O/partition_probe.txt 2>&1; grep -v "^\[bench\]\|amdgpu.ids" $O/partition_probe.txt | tail -20
timeout -s KILL 1800 python -m pytest tests/test_gpu_p2p.py tests/test_gpu_multi_device.py tests/test_gpu_dist.py tests/test_gpu_setup.py -m gpu -q --tb=short --durations=8 -x 2>&1 | tail -40 > $O/pytest_gpu_summary.txt
tail -22 $O/pytest_gpu_summary.txt
