#!/bin/bash
# full GPU check: every -m gpu test, then the contract bench line.  usage: gpu_full.sh <tag>
tag=${1:-full}; O=gpurun_out/$tag; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "rc=$?" >> $O/gpu_tests.log; tail -4 $O/gpu_tests.log
timeout 400 python bench.py --no-pmc > $O/bench_C2.json 2> $O/bench_C2.err; tail -2 $O/bench_C2.err
python - <<PY
import json
d=json.load(open("$O/bench_C2.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],3), "roofline.frac", round(d["roofline"]["frac"],4), "matcher ms", round(d["roofline"]["avg_launch_ms"],4), "conv TF", round(d["roofline_conv"]["achieved"],1), "conv ms/step", round(d["roofline_conv"]["all_conv_layers_ms_per_step"],3))
print("cpu_baseline", d.get("cpu_baseline"))
PY
