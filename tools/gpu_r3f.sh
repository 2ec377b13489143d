#!/bin/bash
mkdir -p gpurun_out/r3f
python tools/ablate.py C2 64 split > gpurun_out/r3f/ablate_C2_split.txt 2>&1
python tools/ablate.py C2 64 > gpurun_out/r3f/ablate_C2_nchw.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_golden_r2.py -x -q > gpurun_out/r3f/pytest_fast.txt 2>&1
bash tools/pmc_v3.sh v3 0 > gpurun_out/r3f/pmc_v3.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3f/ablate_C2_split.txt; grep -v amdgpu.ids gpurun_out/r3f/ablate_C2_nchw.txt | head -3; tail -4 gpurun_out/r3f/pytest_fast.txt; grep -A22 "^TAG" gpurun_out/r3f/pmc_v3.txt
