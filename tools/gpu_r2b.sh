#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -q -m gpu > $O/pytest_fast.log 2>&1; grep -E "passed|failed|FAILED|Error" $O/pytest_fast.log | head -30; grep "parity" $O/pytest_fast.log | cut -c1-330 | head -60
timeout 300 python tools/ablate.py C2 64 > $O/ablate.log 2>&1
timeout 300 python tools/ablate.py C2 64 split >> $O/ablate.log 2>&1
cat $O/ablate.log
