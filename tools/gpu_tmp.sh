#!/bin/bash
# scratch GPU session: persistent conv A/B
cd $GRAFT_REPO_ROOT; O=gpurun_out/persist; mkdir -p $O
{
echo "== persistent (default) =="; CONV_DEV_LIB=1 python tools/conv_kscale.py
echo "== one workgroup per tile (MAGNET_CONV_VARIANT=4096) =="; CONV_DEV_LIB=1 MAGNET_CONV_VARIANT=4096 python tools/conv_kscale.py
echo "== stacks, persistent =="; CONV_DEV_LIB=1 python tools/bench_conv_mx.py
echo "== stacks, per tile =="; CONV_DEV_LIB=1 MAGNET_CONV_VARIANT=4096 python tools/bench_conv_mx.py
} 2>&1 | grep -v amdgpu.ids | tee $O/ab.log
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -5 | tee $O/tests.log
