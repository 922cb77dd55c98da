R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03h; mkdir -p $O; cd $R
cd /tmp; export TMPDIR=/tmp
for E in 0 1; do
  if [ $E = 1 ]; then export GMG_EXP_COALESCE=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$E -- python $R/bench.py --steps 20 --warmup 3 --cpu-cycles 0 --no-variants > /dev/null 2> $O/prof.log
  python - <<PY
import csv,glob
f=glob.glob("$O/prof$E/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "delta_ep" in r["Name"] or "gs_block_ep" in r["Name"]: print("exp=$E", r["Name"][:40], r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
  rm -rf $O/prof$E
done
