#!/bin/bash
out=gpurun_out/r06symv; mkdir -p $out
python scripts/symv_sweep.py 2>&1 | grep "^rows" | tee $out/sweep.txt
for R in 1 2 4; do for U in 2 4 8; do GMG_SYMV_ROWS=$R GMG_SYMV_STRIDES=$U python scripts/symv_sweep.py 2>&1 | grep "^rows" | tee -a $out/sweep.txt; done; done
