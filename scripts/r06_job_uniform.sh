#!/bin/bash
out=gpurun_out/r06uni; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_setup.py -m gpu -x -q -k "equally_wide or 16_bit" 2>&1 | tail -15 > $out/pytest.txt
cat $out/pytest.txt
python scripts/uniform_ab.py 2>&1 | grep -v amdgpu.ids | tee $out/uniform_ab.txt
