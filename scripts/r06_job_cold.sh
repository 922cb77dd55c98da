#!/bin/bash
out=gpurun_out/r06cold; mkdir -p $out
timeout 1200 python -m pytest tests/test_gpu_setup.py -m gpu -x -q 2>&1 | tail -8 > $out/pytest_setup.txt
cat $out/pytest_setup.txt
python scripts/cold_setup_marks.py natural 2>&1 | grep -v amdgpu.ids | tee $out/cold_marks_natural.txt
python scripts/cold_setup_marks.py random 2>&1 | grep -v amdgpu.ids | tee $out/cold_marks_random.txt
