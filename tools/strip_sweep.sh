O=gpurun_out/r5; mkdir -p $O
for wl in C4L C2Lf C2L; do for s in 0 1 2 4 8; do
  echo -n "$wl strip=$s " >> $O/strip_sweep.log
  MAGNET_STRIP=$s timeout 120 python bench.py --dev-lib --kernel-only --workload $wl --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  %.1f %%' % (d['ms_per_step'], 100*d['roofline']['frac']))" >> $O/strip_sweep.log
done; done
cat $O/strip_sweep.log
