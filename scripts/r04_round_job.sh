# Evidence run of round 4 (one gpurun call): bench line, rocprofv3 stats / timelines for the bench workload and the other configs, size scaling,
# set-up breakdowns.  The GPU suite and PMC passes are separate calls (scripts/r04_pmc.sh).  Usage: bash scripts/r04_round_job.sh <tag>
TAG=${1:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04$TAG; mkdir -p $O
cd $R
timeout -s KILL 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.json; echo
timeout -s KILL 300 python bench.py --n1 2829 --n2 2829 --cpu-cycles 0 --no-variants > $O/bench_8m.json 2> $O/bench_8m.err
timeout -s KILL 400 python bench.py --n1 4483 --n2 4483 --cpu-cycles 0 --no-variants > $O/bench_20m.json 2> $O/bench_20m.err
bash scripts/r04_prof.sh $TAG 3m 3m_random:--config:4r pointcloud:--config:3 3m_smoothing_d3:--config:4s 3m_bilaplacian:--config:5b 722k:--config:2 > /dev/null 2>&1
cd $R
for S in 2 1; do GMG_DIST_BACKEND=gloo timeout -s KILL 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 2 --shard-levels $S 2>$O/dist_shard$S.err | tail -1 > $O/bench_2ranks_1gpu_shard$S.json; done
timeout -s KILL 200 python scripts/setup_breakdown.py 2>&1 | grep -A3 "^set_system" > $O/setup_breakdown_natural.txt
timeout -s KILL 200 python scripts/setup_breakdown.py random 2>&1 | grep -A3 "^set_system" > $O/setup_breakdown_random.txt
timeout -s KILL 200 python scripts/hierarchy_timing.py 2>&1 | tail -2 > $O/hierarchy_timing.txt
find $O -name "*.err" -size -1k -delete; du -sh $O; ls $O
