#!/bin/bash
# what the host's wait between two cycles costs: the timed loop with and without it (GMG_EXPERIMENT_NOWAIT: measurement only)
out=gpurun_out/r06gap; mkdir -p $out
for rep in 1 2; do
python bench.py --steps 200 --warmup 20 --cpu-cycles 0 --no-variants > $out/wait_$rep.json 2> $out/wait_$rep.err
GMG_EXPERIMENT_NOWAIT=1 python bench.py --steps 200 --warmup 20 --cpu-cycles 0 --no-variants > $out/nowait_$rep.json 2> $out/nowait_$rep.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r06gap/*.json')):
    try: print(f, json.load(open(f))['ms_per_step'])
    except Exception as e: print(f, 'failed', e)
PY
