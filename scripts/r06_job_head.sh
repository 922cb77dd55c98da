#!/bin/bash
# the head of the next cycle: its test, the parity file, and the bench line with and without it
out=gpurun_out/r06head; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "head_of_the_next or restriction_fused or device_built" 2>&1 | tail -15 > $out/pytest_head.txt
cat $out/pytest_head.txt
for rep in 1 2; do
python bench.py --steps 200 --warmup 20 --cpu-cycles 0 --no-variants > $out/head_$rep.json 2> $out/head_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06head/*.json')):
    try:
        j=json.load(open(f)); print(f, j['ms_per_step'], j.get('solve_ms'), j.get('iterations_to_1e-4'))
    except Exception as e: print(f, 'failed', e)
PY
