# round 5, job b: timeline of a cold gmg_set_system (kernels + copies + HIP API calls)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r05b; mkdir -p $O
cd $R
for ORD in natural; do
  GMG_TRACE=setup timeout -s KILL 300 python scripts/setup_trace.py $ORD > $O/setup_trace_$ORD.txt 2>&1
  (cd /tmp; export TMPDIR=/tmp; timeout -s KILL 400 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $O/prof_$ORD -- python $R/scripts/setup_trace.py $ORD > $O/prof_$ORD.log 2>&1)
  K=$(ls $O/prof_$ORD/*/*kernel_trace.csv | head -1); M=$(ls $O/prof_$ORD/*/*memory_copy_trace.csv | head -1)
  python scripts/setup_timeline.py $K $M > $O/setup_timeline_$ORD.txt 2>&1
  grep SETUP_TRACE $O/prof_$ORD.log >> $O/setup_timeline_$ORD.txt
  A=$(ls $O/prof_$ORD/*/*hip_api_trace.csv | head -1)
  python - "$A" "$K" > $O/hip_api_$ORD.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted(int(r["Start_Timestamp"]) for r in csv.DictReader(open(sys.argv[2])) if "rap_rows<0>" in r["Kernel_Name"])
# last cold set-up: last group of rap_rows<0> launches
starts = [ks[0]]
for t in ks[1:]:
    if t - starts[-1] > 50e6: starts.append(t)
t_rap = starts[-1]
rows = [r for r in rows if t_rap - 40e6 < int(r["Start_Timestamp"]) < t_rap + 60e6]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if d > 50e3:
        print(f'{(int(r["Start_Timestamp"]) - t0) / 1e6:8.2f} ms  {d / 1e3:9.1f} us  tid {r.get("Thread_Id", "?")}  {r["Function"]}')
PY
  rm -rf $O/prof_$ORD
done
ls -la $O
