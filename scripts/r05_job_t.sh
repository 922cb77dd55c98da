R=$GRAFT_REPO_ROOT; cd $R
timeout -s KILL 1500 python -m pytest tests/test_gpu_setup.py -m gpu -q --tb=short -x -k "random_sequences or another_pattern or same_pattern" 2>&1 | tail -15
