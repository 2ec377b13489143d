#!/bin/bash
# round 4: quick matcher iteration — parity of the v4-routed shapes, A/B timing, counters.  usage: gpu_session_b.sh <tag> [pmc]
tag=${1:-r4b}; O=gpurun_out/$tag; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fast_matcher.py -x -q -m gpu -k "bf16 or sweep or tiny" > $O/fast_matcher.log 2>&1; echo "rc=$?" >> $O/fast_matcher.log
tail -3 $O/fast_matcher.log
ABLATE_SHORT=1 timeout 300 python tools/ablate.py C2 64 split 2>/dev/null | tee $O/ablate_C2_split.log
if [ "$2" = "pmc" ]; then bash tools/pmc_v3.sh $tag 0 --no-pmc --sustain-s 0 2>&1 | grep -v "^  GRBM\|SQ_BUSY\|SQ_WAVES " | tee $O/pmc.txt; fi
