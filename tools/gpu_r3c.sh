#!/bin/bash
mkdir -p gpurun_out/r3c
./tools/ubench/issue_rate 2>&1 | grep -E "^s_|^v_fma \+|^# cyc" > gpurun_out/r3c/salu_rate.txt
bash tools/pmc_v3.sh v3 0 > gpurun_out/r3c/pmc_v3.txt 2>&1
bash tools/pmc_v3.sh r2 $((4 | (0x100 << 8))) > gpurun_out/r3c/pmc_r2.txt 2>&1
cat gpurun_out/r3c/salu_rate.txt; tail -30 gpurun_out/r3c/pmc_v3.txt
