#!/bin/bash
# GPU box: kernel-only matcher bench over the BASELINE.json configs (one JSON line each).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/configs.jsonl; : > $out
for wl in C1 C2 C4 C5 C2L C2Lf C4L shipped; do
  timeout 300 python bench.py --kernel-only --workload $wl --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 >> $out
done
python - <<PY
import json
for l in open("$out"):
    d=json.loads(l); r=d["roofline"]; c=d["config"]
    print(f"{c['workload'][:70]:70s} B={c['frames_per_gpu_per_step']:3d} {r['avg_launch_ms']:8.3f} ms  {r['achieved']:7.1f} GB/s  frac {r['frac']:.3f}  {d['cost_volume_frames_per_s']:9.1f} frames/s")
PY
