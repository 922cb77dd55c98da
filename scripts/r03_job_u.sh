R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_p2p.py tests/test_dropin_api.py -m gpu -q -x 2>&1 | tail -25
