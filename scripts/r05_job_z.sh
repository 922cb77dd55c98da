R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05z; mkdir -p $O; cd $R
timeout -s KILL 900 python scripts/soak.py 1500 2>&1 | tail -2
timeout -s KILL 2400 python scripts/run_configs.py --configs 1 2 3 4 4r 5 6 --oracle-max-n 1100000 > $O/configs_full_size.json 2> $O/configs.err; tail -c 600 $O/configs_full_size.json; echo; tail -3 $O/configs.err
