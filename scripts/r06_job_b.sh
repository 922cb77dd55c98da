R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
timeout -s KILL 600 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -x -k "device" 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -30 > $O/pytest_inv.txt; tail -5 $O/pytest_inv.txt
timeout -s KILL 600 python scripts/coarse_inverse_timing.py > $O/coarse_timing.txt 2>&1; cat $O/coarse_timing.txt | grep -v amdgpu.ids
for W in 2 4 8; do timeout -s KILL 300 python scripts/p2p_hybrid_probe.py $W poisson-big 2 2>&1 | grep "exact\|hybrid" > $O/hybrid_$W.txt; cat $O/hybrid_$W.txt; done
