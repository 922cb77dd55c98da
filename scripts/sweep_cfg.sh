#!/bin/bash
# usage: scripts/sweep_cfg.sh "<bench args 1>" "<bench args 2>" ...   -> one summary line per configuration
for a in "$@"; do
  python bench.py --steps 20 --warmup 3 --cpu-cycles 0 $a 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read())
k=j['roofline']['other_fine_kernels']
print('$a', '| ms/cycle %.3f iters %d solve_ms %.1f | fineGS %.1f us/launch %.0f GB/s | res %.0f restrict %.0f prol %.0f norm %.0f us' % (j['value'], j['iterations_to_1e-4'], j['solve_ms'], 1e3*j['roofline']['launch_ms'], j['roofline']['achieved'], 1e3*k['residual']['ms'],1e3*k['restrict']['ms'],1e3*k['prolong_add']['ms'],1e3*k['norm']['ms']))
"
done
