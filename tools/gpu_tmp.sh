#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/hotbox; mkdir -p $O
b() { python bench.py --no-pmc --no-cpu-baseline --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],1), 'frames/s', round(d['ms_per_step'],3), 'ms  matcher', round(d['roofline']['avg_launch_ms'],4), ' conv TF', round(d['roofline_conv']['achieved'],1))"; }
{
rocm-smi --showtemp --showpower --showclocks 2>/dev/null | grep -i "temp\|power\|sclk" | head -8
b fresh1; b fresh2
timeout 300 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_conv.py -q -m gpu 2>&1 | tail -1
rocm-smi --showtemp --showpower 2>/dev/null | grep -i "temp\|power" | head -6
b after_tests1; b after_tests2
python bench.py --no-pmc --no-cpu-baseline --sustain-s 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('with 20 s sustained', round(d['value'],1), round(d['sustained_frames_per_s'],1))"
rocm-smi --showtemp --showpower 2>/dev/null | grep -i "temp\|power" | head -6
b after_sustain
} 2>&1 | tee $O/hotbox.log
