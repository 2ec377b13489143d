O=gpurun_out/r4n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -3 | tee $O/conv_tests.log
timeout 300 python tools/bench_conv_mx.py 64 2>/dev/null | tee $O/bench_conv.log
echo "== round-3 loop (dev variant 512)"; CONV_DEV_LIB=1 MAGNET_CONV_VARIANT=512 timeout 300 python tools/bench_conv_mx.py 64 2>/dev/null | tee -a $O/bench_conv.log
