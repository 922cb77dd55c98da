R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_setup.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -4
python scripts/setup_breakdown.py 2>&1 | grep -A3 "set_system" | head -12
python scripts/setup_breakdown.py random 2>&1 | grep -A3 "set_system" | head -8
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python $R/bench.py --steps 5 --warmup 2 --cpu-cycles 0 --no-variants > /dev/null 2> $O/prof.log
python - <<PY
import csv,glob
f=glob.glob("$O/prof/*/*kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "gmgs::" in r["Name"] or "gmgh::" in r["Name"]: print(r["Name"][:60], r["Calls"], "total_us", round(float(r["TotalDurationNs"])/1e3,1), "max_us", round(float(r["MaxNs"])/1e3,1))
PY
rm -rf $O/prof
