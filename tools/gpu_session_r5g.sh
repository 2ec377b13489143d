#!/bin/bash
O=gpurun_out/r5g; mkdir -p $O
for s in 0 1 0 1; do
  echo -n "C2 v3 MAGNET_STRIP=$s " >> $O/v3_strip.log
  MAGNET_STRIP=$s timeout 120 python bench.py --dev-lib --kernel-only --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  %.2f %%' % (d['ms_per_step'], 100*d['roofline']['frac']))" >> $O/v3_strip.log
done
for wl in C4 C5; do for s in 0 1; do
  echo -n "$wl v3 MAGNET_STRIP=$s " >> $O/v3_strip.log
  MAGNET_STRIP=$s timeout 120 python bench.py --dev-lib --kernel-only --workload $wl --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  %.2f %%' % (d['ms_per_step'], 100*d['roofline']['frac']))" >> $O/v3_strip.log
done; done
cat $O/v3_strip.log
