R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03v; mkdir -p $O; cd $R
python scripts/ldlt_team_bench.py scripts/micro/coarse_4.npz scripts/micro/coarse_4r.npz scripts/micro/coarse_3.npz 2>&1 | grep -E "n=|4 threads|8 threads|1 thread" | cut -c1-250 | tee $O/ldlt_chains.txt
( python scripts/ab_cycle.py --config 4r --label "4r"; python scripts/ab_cycle.py --config 3 --label "3"; python scripts/ab_cycle.py --config 4 --label "4";  python scripts/ab_cycle.py --config 4s --label "4s" ) 2>/dev/null | tee $O/ab_chains.jsonl
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cycle_model.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -2
