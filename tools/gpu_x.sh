#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/x; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "fast_matcher or golden or parity" 2>&1 | tail -3
for cfg in "C4" "C2 --feat-dtype fp32" "C5 --feat-dtype fp32"; do for f in 0x8000 0 0x8000 0; do
  MAGNET_DEV_FLAGS=$f python bench.py --dev-lib --kernel-only --workload $cfg --steps 100 --warmup 150 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | tail -1 > $O/_k.json
  python -c "
import json; d=json.load(open('gpurun_out/x/_k.json')); print('$cfg flags $f: %.3f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
done; done
