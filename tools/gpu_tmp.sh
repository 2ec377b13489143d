#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/persist; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_fnet.py -x -q -m gpu 2>&1 | tail -3 | tee $O/tests_epi3.log
{
echo "== staged epilogue, bias hoisted =="; python tools/conv_kscale.py
python tools/bench_fnet.py --frames 8 --skip-torch
python tools/bench_fnet.py --frames 8 --skip-torch
python tools/bench_conv_mx.py
} 2>&1 | grep -v amdgpu.ids | tee $O/epi3.log
