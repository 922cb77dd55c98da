cd $GRAFT_REPO_ROOT
for spec in "2 3000 poisson-big 2" "3 1500 poisson-big 2" "4 1000 smoothing-d3 2" "4 1000 poisson-big 1" "8 150 poisson 2"; do
timeout -s KILL 900 python scripts/soak_p2p.py $spec 2>&1 | grep -v "Gloo\|amdgpu.ids" | tail -2
done
