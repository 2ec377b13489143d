#!/bin/bash
# round 6, session b (GPU box): records that close the review's items 2, 4a, 7 + the launch-quantisation probe -> gpurun_out/r6b/
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc --sustain-s 0"
# (1) tiles per launch vs CUs: 256-row tiles over 256 CUs, one workgroup per CU -> the launch time should step with ceil(tiles / 256)
: > $O/frames_sweep.jsonl
for f in 60 62 63 64 66 68 64; do timeout 120 $B --frames $f 2>/dev/null | tail -1 >> $O/frames_sweep.jsonl; done
# (2) mask head on a side stream beside matcher + G-Net (item 2): same-box A/B, then the kernel trace of the overlapped form
: > $O/overlap_ab.jsonl
for i in 1 2; do timeout 120 $B 2>/dev/null | tail -1 >> $O/overlap_ab.jsonl; timeout 120 $B --overlap 2>/dev/null | tail -1 >> $O/overlap_ab.jsonl; done
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/overlap_trace -o t -- $B --overlap --steps 6 --warmup 3 > $O/overlap_trace.log 2>&1
# (3) the matcher's clock behind matrix-core work, by idle gap
timeout 300 python tools/clock_recovery.py 2>&1 | grep -v amdgpu.ids > $O/clock_recovery.jsonl
# (4) what the quad-form (mu, sigma) bytes cost the production matcher (timing only)
ABLATE_HALFQ=1 timeout 200 python tools/ablate.py C2 64 split 2>&1 | grep -v amdgpu.ids > $O/ablate_halfq.log
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
for f in $(find $O -name "*kernel_trace.csv"); do python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_mfma", "cv_v3", "pack_", "upsample"))][-40:]
with open(sys.argv[1] + ".tail.txt", "w") as f:
    for r in keep:
        f.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:12.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:90]}\n')
PY
rm $f; done
cat $O/clock_recovery.jsonl; cat $O/ablate_halfq.log; python - <<'PY'
import json
for f in ("gpurun_out/r6b/frames_sweep.jsonl", "gpurun_out/r6b/overlap_ab.jsonl"):
    for l in open(f):
        try: d = json.loads(l)
        except Exception: continue
        print(f.split("/")[-1], d["config"]["frames_per_gpu_per_step"], round(d["value"]), round(d["ms_per_step"], 3), round(d["roofline"]["avg_launch_ms"], 4), round(d.get("roofline_conv", {}).get("all_conv_layers_ms_per_step", 0), 3))
PY
