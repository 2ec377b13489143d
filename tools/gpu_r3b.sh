#!/bin/bash
# round 3, session b: first run of the round-3 matcher (dev build): kernel-only A/B against the round-2 kernels + parity tests
mkdir -p gpurun_out/r3b
python tools/ablate.py C2 64 split > gpurun_out/r3b/ablate_C2_split.txt 2>&1
python tools/ablate.py C2 64 > gpurun_out/r3b/ablate_C2_nchw.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_golden_r2.py -x -q > gpurun_out/r3b/pytest_fast.txt 2>&1
tail -8 gpurun_out/r3b/ablate_C2_split.txt gpurun_out/r3b/ablate_C2_nchw.txt; tail -15 gpurun_out/r3b/pytest_fast.txt
