O=$GRAFT_REPO_ROOT/gpurun_out/r04l; mkdir -p $O
C="torus48x40 torus96x80 cloud3000 cloud6000 cloud20000 cloud120000"
timeout -s KILL 500 python scripts/fine_blocks_probe.py $C > $O/fine_blocks_probe.jsonl 2>$O/probe.err
GMG_FINE_BLOCKS_GROWN=1 timeout -s KILL 500 python scripts/fine_blocks_probe.py $C 2>>$O/probe.err | grep block_from_level >> $O/fine_blocks_probe.jsonl
cat $O/fine_blocks_probe.jsonl
timeout -s KILL 900 python -m pytest tests/test_gpu_fine_blocks.py tests/test_gpu_parity.py tests/test_gpu_setup.py -m gpu -q -x --tb=short 2>&1 | tail -60 > $O/pytest_third.txt
tail -40 $O/pytest_third.txt
