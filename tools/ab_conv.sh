#!/bin/bash
# GPU box: same-box A/B of the convolution K-loop variants (MAGNET_CONV_VARIANT bits, tools/README.md) inside the C2 step.
# Needs the dev library (python -m magnet_amd.build --dev): the product build ignores the switch.
# Two passes over the list so that drift shows.  -> gpurun_out/ab_conv_variants.txt (copy to profiles/<round>/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/ab_conv_variants.txt; : > $O
for pass in 1 2; do
for v in 0 16 8 1 2 64 128 32; do
  MAGNET_CONV_VARIANT=$v timeout 120 python bench.py --dev-lib --no-pmc --no-cpu-baseline --sustain-s 0 2>/dev/null | tail -1 > gpurun_out/_ab.json
  python - >> $O <<PY
import json
d=json.load(open("gpurun_out/_ab.json"))
names={0:"default: 8-wave ping-pong x register window, column-owned tail",16:"4-wave register window + 3-slot weight ring",8:"4-wave 2-slot LDS window (flat loop)",1:"one A stage per tap (round-1 loop, VGPR-form MFMAs)",2:"8-wave ping-pong, one A stage per tap",64:"8-wave ping-pong x 2-slot LDS window",128:"8-wave ping-pong x LDS window x fragment double-buffering",32:"default loop, row-owned tail"}
print("pass $pass variant %3d  step %.3f ms  convs %.3f ms  %.1f TF   %s" % ($v, d["ms_per_step"], d["roofline_conv"]["all_conv_layers_ms_per_step"], d["roofline_conv"]["achieved"], names[$v]))
PY
done; done
cat $O
