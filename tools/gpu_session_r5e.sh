#!/bin/bash
# round 5: the fused-tail convolution kernels on v_mfma_f32_32x32x16_bf16 (M32) — parity, then same-box A/B against the 16x16x32 form
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -s > $O/conv_tests.log 2>&1; echo "rc=$?" >> $O/conv_tests.log; tail -n 4 $O/conv_tests.log
for v in 0 16384 0 16384; do
  echo "== MAGNET_CONV_VARIANT=$v" >> $O/conv_m32_ab.log
  MAGNET_CONV_VARIANT=$v timeout 200 python bench.py --dev-lib --no-cpu-baseline --no-pmc --sustain-s 3 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); c=d['roofline_conv']
print('step %.3f ms  %.0f frames/s  sustained %.0f  conv %.3f ms/step  %.1f TF  3x3 launch %.3f ms  matcher %.3f ms' % (d['ms_per_step'], d['value'], d['sustained_frames_per_s'], c['all_conv_layers_ms_per_step'], c['achieved'], c['avg_launch_ms'], d['roofline']['avg_launch_ms']))" >> $O/conv_m32_ab.log
done
cat $O/conv_m32_ab.log
