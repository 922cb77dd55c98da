R=$GRAFT_REPO_ROOT; cd $R
for i in 1 2 3; do timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -1; done
for i in 1 2 3; do timeout 600 python scripts/parity_sweep.py 2>/dev/null | tail -1 | cut -c1-60; done
timeout 900 python scripts/soak.py 2>&1 | tail -1 | cut -c1-80
