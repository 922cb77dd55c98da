# evidence of the round's last build, one job: suite, bench line, kernel stats + timelines, PMC passes, ranks on one GPU.  bash scripts/r06_evidence.sh <tag>
T=${1:-z}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06$T; mkdir -p $O; cd $R
cp profiles/r06/commit.txt $O/commit.txt
timeout -s KILL 2700 python -m pytest tests -m gpu -q --tb=short --durations=8 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -40 > $O/pytest_gpu_summary.txt
grep -n "passed\|failed\|^FAILED" $O/pytest_gpu_summary.txt | head
timeout -s KILL 1200 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 200 $O/bench.json; echo
bash scripts/r06_prof.sh $T 3m 3m_smoothing_d3:--config:4s 722k:--config:2 pointcloud:--config:3 3m_bilaplacian:--config:5b 3m_random:--config:4r > $O/prof.log 2>&1
bash scripts/r06_pmc.sh $T 3m; bash scripts/r06_pmc.sh $T 3m_smoothing_d3 --config 4s; bash scripts/r06_pmc.sh $T 722k --config 2
cd $R
python scripts/coarse_inverse_timing.py 2>&1 | grep -v amdgpu > $O/coarse_inverse_timing.txt
python scripts/head_ab.py 2>&1 | grep -v amdgpu > $O/head_of_next_cycle_ab.txt
python scripts/uniform_ab.py 2>&1 | grep -v amdgpu > $O/uniform_slices_ab.txt
(python scripts/cold_setup_marks.py natural; python scripts/cold_setup_marks.py random) 2>&1 | grep -v amdgpu > $O/cold_setup_marks.txt
for W in 2 4 8; do timeout -s KILL 300 python scripts/p2p_hybrid_probe.py $W poisson-big 2 2>&1 | grep "exact\|hybrid"; done > $O/hybrid_smoother_ranks_on_one_gpu.txt
timeout -s KILL 900 python bench.py --n1 2828 --n2 2828 --no-variants --cpu-cycles 0 > $O/bench_8m.json 2> $O/bench_8m.err; tail -c 150 $O/bench_8m.json; echo
timeout -s KILL 1500 python bench.py --n1 4472 --n2 4472 --no-variants --cpu-cycles 0 > $O/bench_20m.json 2> $O/bench_20m.err; tail -c 150 $O/bench_20m.json; echo
for W in 2 4; do GMG_DIST_BACKEND=gloo timeout -s KILL 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $W --master-addr 127.0.0.1 --master-port 2953$W bench.py --gpus $W --steps 10 --warmup 2 2>$O/dist_$W.err | tail -1 > $O/bench_${W}ranks_1gpu.json; tail -c 300 $O/bench_${W}ranks_1gpu.json; echo; done
