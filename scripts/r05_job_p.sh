R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05p; mkdir -p $O
cd $R
timeout -s KILL 1500 python -m pytest tests/test_gpu_multi_device.py -m gpu -q --tb=short -x 2>&1 | tail -15
