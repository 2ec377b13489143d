O=gpurun_out/r4e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -x -q -m gpu > $O/fast_matcher.log 2>&1; echo "rc=$?" >> $O/fast_matcher.log; tail -3 $O/fast_matcher.log
ABLATE_SHORT=1 timeout 300 python tools/ablate.py C2 64 split 2>/dev/null | tee $O/ablate_C2_split.log
