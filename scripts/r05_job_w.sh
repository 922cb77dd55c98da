cd $GRAFT_REPO_ROOT
GMG_TRACE=setup timeout -s KILL 300 python scripts/cold_setup_marks.py natural 2>&1 | grep -v "^\[bench\]" | tail -60 | cut -c1-600
