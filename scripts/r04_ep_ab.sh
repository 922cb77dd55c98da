# level-1 sweep A/B over experimental builds (gravo_mg_amd/lib/variants/libgmg_<name>.so): bash scripts/r04_ep_ab.sh name...
O=$GRAFT_REPO_ROOT/gpurun_out/r04b; mkdir -p $O; rm -f $O/ep_ab.txt
for v in "$@"; do echo "== $v" >> $O/ep_ab.txt; GMG_LIB_PATH=$GRAFT_REPO_ROOT/gravo_mg_amd/lib/variants/libgmg_$v.so python scripts/l1_sweep_ab.py "" 2>&1 | grep "L1 sweep" >> $O/ep_ab.txt; done
cat $O/ep_ab.txt
