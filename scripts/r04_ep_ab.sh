# level-1 sweep A/B: persistent workgroups (GMG_EP_WAVES_PER_CU)
O=$GRAFT_REPO_ROOT/gpurun_out/r04b; mkdir -p $O; rm -f $O/ep_ab.txt
run() { echo "== WPC=$1" >> $O/ep_ab.txt; GMG_EP_WAVES_PER_CU=$1 python scripts/l1_sweep_ab.py "" 2>&1 | grep "L1 sweep" >> $O/ep_ab.txt; }
for w in "$@"; do run $w; done
cat $O/ep_ab.txt
