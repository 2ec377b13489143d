#!/bin/bash
O=gpurun_out/r5h; mkdir -p $O
run() { echo -n "$1 MINW=$2 VG=$3: " >> $O/f64_occ.log
  MAGNET_F64_MINW=$2 MAGNET_F64_VG=$3 timeout 120 python bench.py --dev-lib --kernel-only --workload $1 --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f ms  %.2f %%' % (d['ms_per_step'], 100*d['roofline']['frac']))" >> $O/f64_occ.log; }
run C2L 0 0; run C2L 6 2; run C2L 6 1; run C2L 8 2; run C2L 6 4; run C2L 0 2; run C2L 0 0
timeout 600 python -m pytest tests/test_gpu_fast_matcher.py -q -m gpu 2>&1 | tail -n 2 >> $O/f64_occ.log
cat $O/f64_occ.log
