#!/bin/bash
# round 5, first GPU session: parity of the texel-pair matchers (cost_volume_fast.hip / cost_volume_fast64.hip), same-box A/B against the
# quad items of rounds 2 - 4, the one-rank RCCL group + live binding counters of bench.py
set -x
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -x -q -m gpu -s > $O/fast_matcher.log 2>&1; echo "rc=$?" >> $O/fast_matcher.log
tail -n 5 $O/fast_matcher.log
# every D > 32 case through the batched-view kernel (dev flag 0x100 skips cost_volume_v3.hip)
MAGNET_TEST_DEV_LIB=1 MAGNET_DEV_FLAGS=0x100 timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -x -q -m gpu -s -k "baseline_shapes or sweep or ragged or tiny or batch_independence" > $O/fast_matcher_fast64_everywhere.log 2>&1; echo "rc=$?" >> $O/fast_matcher_fast64_everywhere.log
tail -n 5 $O/fast_matcher_fast64_everywhere.log
# ... and through the per-view kernel (0x800 skips fast64 as well)
MAGNET_TEST_DEV_LIB=1 MAGNET_DEV_FLAGS=0x900 timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -x -q -m gpu -s -k "baseline_shapes or sweep or ragged or tiny" > $O/fast_matcher_perview_everywhere.log 2>&1; echo "rc=$?" >> $O/fast_matcher_perview_everywhere.log
tail -n 5 $O/fast_matcher_perview_everywhere.log
for cfg in "C2L 4" "C4L 4" "shipped 64" "C1 64" "C4 16"; do
  set -- $cfg
  ABLATE_TX=1 timeout 300 python tools/ablate.py $1 $2 split >> $O/ablate_tx.log 2>&1
done
ABLATE_TX=1 timeout 300 python tools/ablate.py C2 64 split >> $O/ablate_tx.log 2>&1
cat $O/ablate_tx.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "bench_size" > $O/parity_bench_size.log 2>&1; echo "rc=$?" >> $O/parity_bench_size.log
tail -n 8 $O/parity_bench_size.log
timeout 400 python bench.py > $O/bench_C2.json 2> $O/bench_C2.err
cat $O/bench_C2.json; tail -n 5 $O/bench_C2.err
