#!/bin/bash
mkdir -p gpurun_out/r3d
python tools/ablate.py C2 64 split > gpurun_out/r3d/ablate_C2_split.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_golden_r2.py -x -q > gpurun_out/r3d/pytest_fast.txt 2>&1
bash tools/pmc_v3.sh v3 0 > gpurun_out/r3d/pmc_v3.txt 2>&1
grep -v amdgpu.ids gpurun_out/r3d/ablate_C2_split.txt; tail -5 gpurun_out/r3d/pytest_fast.txt; grep -A22 "^TAG" gpurun_out/r3d/pmc_v3.txt
