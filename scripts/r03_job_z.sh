R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03z; mkdir -p $O; cd $R
for rep in 1 2; do for v in plain bound; do GMG_LIB_PATH=$R/scripts/micro/ab/lib_$v.so python scripts/ab_cycle.py --config 4 --label "$v" 2>/dev/null | cut -c1-200; done; done
for v in plain bound; do GMG_LIB_PATH=$R/scripts/micro/ab/lib_$v.so python scripts/ab_cycle.py --config 4s --label "$v" 2>/dev/null | cut -c1-200; GMG_LIB_PATH=$R/scripts/micro/ab/lib_$v.so python scripts/ab_cycle.py --config 3 --label "$v" 2>/dev/null | cut -c1-200; done
