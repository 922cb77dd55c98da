R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "solve_reaches" 2>&1 | tail -2
