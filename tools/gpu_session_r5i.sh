#!/bin/bash
O=gpurun_out/r5i; mkdir -p $O
MAGNET_TEST_DEV_LIB=1 MAGNET_DEV_FLAGS=0x400 timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -x -q -m gpu > $O/tests_v3tx.log 2>&1; echo "rc=$?" >> $O/tests_v3tx.log; tail -n 12 $O/tests_v3tx.log
for cfg in "C2 64" "C4 24" "C5 30"; do set -- $cfg
  ABLATE_TX=1 timeout 300 python tools/ablate.py $1 $2 split 2>&1 | grep -v amdgpu.ids | head -4 >> $O/ablate_v3tx.log
done
cat $O/ablate_v3tx.log
