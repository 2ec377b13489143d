O=gpurun_out/r4p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | tee $O/tests.log
: > $O/configs.jsonl
for wl in C1 C2 C4 C5 C2L C4L shipped; do
  timeout 120 python bench.py --kernel-only --workload $wl --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | tail -1 >> $O/configs.jsonl
done
python - <<PY
import json
for l in open("$O/configs.jsonl"):
    d=json.loads(l); print(d["config"]["workload"][:24].ljust(26), "frames", d["config"]["frames_per_gpu_per_step"], "ms", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],4))
PY
