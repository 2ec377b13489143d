#!/bin/bash
# round 4, first GPU session: parity of the round-4 matcher + A/B timing against round 3's
set -x
O=gpurun_out/r4a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -x -q -m gpu -s > $O/fast_matcher.log 2>&1; echo "rc=$?" >> $O/fast_matcher.log
tail -5 $O/fast_matcher.log
timeout 600 python -m pytest tests/test_gpu_golden_r2.py tests/test_gpu_golden_r3.py -x -q -m gpu -s > $O/golden.log 2>&1; echo "rc=$?" >> $O/golden.log
tail -5 $O/golden.log
ABLATE_SHORT=1 timeout 300 python tools/ablate.py C2 64 split > $O/ablate_C2_split.log 2>&1
cat $O/ablate_C2_split.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-pmc --no-cpu-baseline --sustain-s 2 > $O/bench_C2.json 2> $O/bench_C2.err
cat $O/bench_C2.json
