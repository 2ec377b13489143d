#!/bin/bash
O=gpurun_out/r5k; mkdir -p $O; rm -f $O/ablate_px2.log
timeout 1200 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_parity.py tests/test_gpu_golden_r2.py tests/test_gpu_golden_r3.py tests/test_gpu_rays_poses.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
for cfg in "C2 64" "C4 24" "C5 30"; do set -- $cfg
ABLATE_PX2=1 timeout 300 python tools/ablate.py $1 $2 split 2>&1 | grep -v amdgpu.ids >> $O/ablate_px2.log
done
cat $O/ablate_px2.log
timeout 300 python bench.py --no-cpu-baseline --no-pmc --sustain-s 2 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.readline()); print('bench: %.0f frames/s  %.3f ms/step  matcher %.4f ms  frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
