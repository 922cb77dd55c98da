R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
for i in 1 2; do python scripts/ldlt_team_bench.py scripts/micro/coarse_4.npz scripts/micro/coarse_4r.npz scripts/micro/coarse_3.npz 2>&1 | grep -E "factorisation" | cut -c1-220; done | tee $O/ldlt.txt
for i in 1 2; do python scripts/setup_breakdown.py random 2>&1 | grep -A2 "^set_system" | head -3 | cut -c1-700; done | tee $O/setup_breakdown_random.txt
python scripts/setup_breakdown.py 2>&1 | grep -A2 "^set_system" | head -3 | cut -c1-700
