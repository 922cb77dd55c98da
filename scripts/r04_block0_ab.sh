# level 0 as a blocked level (block-hybrid sweep, one launch per sweep) against the multicolour sweep (one launch per colour): configs with many colours
O=$GRAFT_REPO_ROOT/gpurun_out/r04i; mkdir -p $O; rm -f $O/block0_ab.txt
for c in 3 5b 4; do for o in "" "block_from_level=0"; do
  timeout -s KILL 300 python scripts/ab_cycle.py --config $c --steps 40 --reps 3 --label "cfg$c $o" $o 2>&1 | tail -1 | cut -c1-330 >> $O/block0_ab.txt
done; done
cat $O/block0_ab.txt
