# round 3, job a: micro-benchmarks (launch floor vs grid barrier), start-of-round bench, kernarg A/B
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03a; mkdir -p $O
cd $R
timeout 120 scripts/micro/grid_barrier.bin > $O/grid_barrier.txt 2>&1; cat $O/grid_barrier.txt
python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
for v in 0 1; do HIP_FORCE_DEV_KERNARG=$v python bench.py --cpu-cycles 0 --no-variants --steps 50 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v', j['ms_per_step'])" ; done | tee $O/kernarg_ab.txt
