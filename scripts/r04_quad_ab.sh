# which blocked levels run the entry-parallel sweep instead of the quad layout (GMG_QUAD_LEVEL_ROWS): cycle time at d = 1 (config 4) and d = 3 (4s)
O=$GRAFT_REPO_ROOT/gpurun_out/r04d; mkdir -p $O; rm -f $O/quad_ab.txt
for q in 131072 32768 4096; do for c in 4 4s 3; do
  GMG_QUAD_LEVEL_ROWS=$q python scripts/ab_cycle.py --config $c --steps 60 --reps 3 --label "quad_rows=$q" 2>/dev/null | tail -1 | cut -c1-260 >> $O/quad_ab.txt
done; done
cat $O/quad_ab.txt
