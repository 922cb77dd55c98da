# level 0 blocked with blocks = runs of 64 points of the hierarchy's cluster order (no breadth-first growth over A, no host copy of A): set-up phases and cycle A/B
O=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $O; rm -f $O/block0_cluster_ab.txt
for c in 3 5b; do for o in "" "block_from_level=0"; do
  GMG_SETUP_TRACE=1 timeout -s KILL 400 python scripts/ab_cycle.py --config $c --steps 40 --reps 3 --label "cfg$c $o" $o > $O/run_${c}_${o%%=*}.log 2>&1
  grep "make_block_ordering\|level 0 \|build_patches" $O/run_${c}_${o%%=*}.log | head -30 >> $O/block0_cluster_ab.txt
  tail -1 $O/run_${c}_${o%%=*}.log | cut -c1-400 >> $O/block0_cluster_ab.txt
done; done
cat $O/block0_cluster_ab.txt
