#!/bin/bash
# round 5, third session: the kernel-only lines of every config (the profile round's `tail -1` had caught RCCL's stdout banner, which
# bench.py now routes to stderr), then the whole GPU suite
O=gpurun_out/r5; mkdir -p $O
: > $O/configs.jsonl
for wl in C1 C2 C4 C5 C2L C2Lf C4L shipped; do
  timeout 120 python bench.py --kernel-only --workload $wl --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | tail -1 >> $O/configs.jsonl
done
timeout 120 python bench.py --kernel-only --nchw-out --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | tail -1 > $O/kernel_only_C2_nchw.json
cut -c1-200 $O/configs.jsonl
timeout 1500 python -m pytest tests -m gpu -q > $O/gpu_tests_full.log 2>&1; echo "rc=$?" >> $O/gpu_tests_full.log
tail -n 5 $O/gpu_tests_full.log
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-pmc --sustain-s 0 > $O/stdout_check.txt 2>/dev/null; wc -l $O/stdout_check.txt
