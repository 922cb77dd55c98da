# full GPU suite (+ durations); argument: output tag
R=$GRAFT_REPO_ROOT; T=${1:-suite}; O=$R/gpurun_out/r06_$T; mkdir -p $O; cd $R
timeout -s KILL 2700 python -m pytest tests -m gpu -q --tb=short --durations=8 2>&1 | grep -v "Gloo\|socket.cpp\|amdgpu.ids" | tail -80 > $O/pytest_gpu_summary.txt
grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_gpu_summary.txt | head -30
