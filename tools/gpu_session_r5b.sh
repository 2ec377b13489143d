#!/bin/bash
# round 5, second session: block order of the batched-view kernel on the full-resolution grids (vertical strips instead of raster order)
set -x
O=gpurun_out/r5b; mkdir -p $O
for strip in 0 4 2 1; do
  echo "== MAGNET_STRIP=$strip" >> $O/strip_order.log
  MAGNET_STRIP=$strip ABLATE_TX=1 ABLATE_SHORT=1 timeout 300 python tools/ablate.py C2L 4 split 2>&1 | grep -v amdgpu.ids >> $O/strip_order.log
done
MAGNET_STRIP=4 ABLATE_TX=1 ABLATE_SHORT=1 timeout 300 python tools/ablate.py C4L 4 split 2>&1 | grep -v amdgpu.ids >> $O/strip_order.log
echo "== smooth inputs (synth smooth=6), raster order" >> $O/strip_order.log
ABLATE_SMOOTH=6 ABLATE_TX=1 ABLATE_SHORT=1 timeout 600 python tools/ablate.py C2L 4 split 2>&1 | grep -v amdgpu.ids >> $O/strip_order.log
cat $O/strip_order.log
