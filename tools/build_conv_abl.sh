#!/bin/bash
# Dev: timing-ablation builds of the convolution kernel's production K loop (conv_mfma.hip, WIN == 2 && PP; CONV_ABL bits: 1 no DMA inside
# the loop, 2 no weight-fragment reads, 4 no window reads, 8 no MFMAs).  Links each against the dev objects of `python -m magnet_amd.build
# --dev` into magnet_amd/libmagnet_hip_abl<bits>.so (git-ignored; travels with gpurun).  Use: CONV_LIB=magnet_amd/libmagnet_hip_abl3.so
# MAGNET_CONV_VARIANT=8192 python tools/conv_kscale.py
cd "$(dirname "$0")/.." || exit 1
python -m magnet_amd.build --dev > /dev/null || exit 1
C=magnet_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fvisibility=hidden -fno-gpu-rdc -Wno-unused-function"
# a non-numeric argument is a macro name for -D (A/B builds of an #ifdef in conv_mfma.hip) -> libmagnet_hip_abl<NAME>.so
for a in "$@"; do
  case $a in [0-9]*) D="-DCONV_ABL=$a";; *) D="-D$a";; esac
  /opt/rocm/bin/hipcc $FLAGS -DMAGNET_DEV $D -c $C/conv_mfma.hip -o $C/conv_mfma.abl$a.o || exit 1
  objs=$(ls $C/*.dev.o | grep -v conv_mfma.dev.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $C/conv_mfma.abl$a.o -o magnet_amd/libmagnet_hip_abl$a.so || exit 1
  echo magnet_amd/libmagnet_hip_abl$a.so
done
