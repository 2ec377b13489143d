#!/bin/bash
# GPU box: the round's official artefacts -> gpurun_out/<tag>/ (tools/collect_profile.py turns them into profiles/<tag>/).
# Every number DESIGN.md quotes must come out of this script (or a file it names under profiles/<tag>/):
#   bench.json            the contract line (python bench.py: live HIP-event roofline + its own rocprofv3 counter passes)
#   stats/                rocprofv3 --kernel-trace --stats of the same command
#   pmc*/                 separate --pmc passes over the same command (FETCH_SIZE, WRITE_SIZE, SQ sets, TA/TCP, TCC)
#   calib_*               FETCH/WRITE calibration on a 1 GiB device copy
#   configs.jsonl         matcher alone (--kernel-only) on every BASELINE config
#   extra/                the step on C3, C4, C5, shipped (D = 5), C2 with packed inputs
#   bench_fnet.json, fnet_layers.txt          row N3 (tools/bench_fnet.py)
#   fvolume_bench.jsonl, fvolume_stats/       row N2 (tools/bench_fvolume.py) + its kernel stats
#   bench_end_to_end.json                     images -> F-Net -> matcher -> G-Net ... (tools/bench_end_to_end.py)
#   ablate_*.log          matcher variants on the dev library (tools/ablate.py), issue_rate / gather_rate microbenchmarks
# Every profiler pass has a short timeout: some TA/TCP/TD counter sets hang rocprofv3 on this pool.  The --pmc passes run the
# contract's own 5 + 20 steps, so that the clock / busy counters describe the chip state the bench line was measured in.
tag=${1:-r6}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$tag; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python bench.py --no-cpu-baseline --no-pmc --sustain-s 0 > $O/stats.log 2>&1
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_MFMA" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_SCA" \
            "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc$i -o p -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-pmc --sustain-s 0 > $O/pmc$i.log 2>&1 || echo "pmc pass $i failed"
done
# the matcher alone and warm: the clock it gets when the matrix-core kernels are not around (300 warm-up launches first)
timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TA_BUSY_avr --output-format csv -d $O/pmc_alone -o p -- python bench.py --kernel-only --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 > $O/pmc_alone.log 2>&1 || echo "pmc alone failed"
for c in FETCH_SIZE WRITE_SIZE; do
timeout 90 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/calib_$c -o c -- python -c "
import torch
x=torch.empty(1<<28,dtype=torch.float32,device='cuda'); y=torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()" > $O/calib_$c.log 2>&1
done
: > $O/configs.jsonl
for wl in C1 C2 C4 C5 C2L C2Lf C4L shipped; do
  # warm: the chip's clock settles over the first ~0.3 s of load (10-launch samples read 8 % slow)
  timeout 120 python bench.py --kernel-only --workload $wl --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | tail -1 >> $O/configs.jsonl
done
timeout 120 python bench.py --kernel-only --nchw-out --steps 200 --warmup 300 --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | tail -1 > $O/kernel_only_C2_nchw.json
mkdir -p $O/extra
timeout 120 python bench.py --no-cpu-baseline --no-pmc --packed-inputs > $O/extra/bench_C2_packed_inputs.json 2>/dev/null
for wl in C3 C4 C5 shipped; do timeout 120 python bench.py --no-cpu-baseline --no-pmc --workload $wl > $O/extra/bench_$wl.json 2>/dev/null; done
for wl in C2 C5; do timeout 200 python bench.py --no-cpu-baseline --no-pmc --with-fnet --workload $wl --steps 5 --warmup 2 --sustain-s 0 > $O/extra/bench_${wl}_with_fnet.json 2>/dev/null; done
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_shipped -o s -- python bench.py --workload shipped --no-cpu-baseline --no-pmc --sustain-s 0 > $O/stats_shipped.log 2>&1
# parity statistics printed by the GPU tests (gate flips, error distributions, loop abs_rel): the numbers DESIGN.md section 2 quotes
timeout 600 python -m pytest tests -m gpu -q -s -k "fast_matcher or golden or parity or rays_poses" 2>&1 | grep -v amdgpu.ids > $O/parity_stats_gpu_tests.txt
# rows N3 / N2 / the whole forward
timeout 200 python tools/bench_fnet.py --frames 8 > $O/bench_fnet.json 2> $O/bench_fnet.err
timeout 200 python tools/bench_fnet.py --frames 8 --skip-torch --profile-layers > $O/fnet_layers.txt 2>&1
timeout 200 python tools/bench_fvolume.py > $O/fvolume_bench.jsonl 2> $O/fvolume.err
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fvolume_stats -o s -- python tools/bench_fvolume.py --steps 5 > $O/fvolume_stats.log 2>&1
timeout 200 python tools/bench_end_to_end.py > $O/bench_end_to_end.json 2> $O/bench_end_to_end.err
timeout 200 python tools/bench_pipeline.py 2>/dev/null | tail -1 > $O/bench_pipeline.json
# matcher variants (dev library) + instruction / gather microbenchmarks
ABLATE_PX2=1 timeout 150 python tools/ablate.py C2 64 split 2>&1 | grep -v amdgpu.ids > $O/ablate_C2_split.log
# round 6: the records that close the review's items (same box as the bench line above)
ABLATE_HALFQ=1 timeout 200 python tools/ablate.py C2 64 split 2>&1 | grep -v amdgpu.ids > $O/ablate_halfq.log           # (mu, sigma) map bytes: fp16 map A/B
timeout 300 python tools/clock_recovery.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|model = MAGNET" > $O/clock_recovery.jsonl      # matcher time vs idle gap behind matrix-core work
# mask head on a side stream (same-box A/B + the kernel trace that shows the kernels serialise), frames-per-step sweep (launch quantisation: 256-row tiles over 256 CUs)
B6="python bench.py --no-cpu-baseline --no-pmc --sustain-s 0 --no-graph"
: > $O/overlap_ab.jsonl; for i in 1 2; do timeout 120 $B6 2>/dev/null | tail -1 >> $O/overlap_ab.jsonl; timeout 120 $B6 --overlap 2>/dev/null | tail -1 >> $O/overlap_ab.jsonl; done
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/overlap_trace -o t -- $B6 --overlap --steps 6 --warmup 3 > $O/overlap_trace.log 2>&1
: > $O/frames_sweep.jsonl; for f in 60 62 63 64 66 68 64; do timeout 120 $B6 --frames $f 2>/dev/null | tail -1 >> $O/frames_sweep.jsonl; done
: > $O/graph_replay_ab.jsonl; for i in 1 2 3; do timeout 120 python bench.py --no-cpu-baseline --no-pmc --sustain-s 0 2>/dev/null | tail -1 >> $O/graph_replay_ab.jsonl; timeout 120 $B6 2>/dev/null | tail -1 >> $O/graph_replay_ab.jsonl; done
timeout 300 python tools/wino_bound.py 2>&1 | grep -v amdgpu.ids > $O/wino_bound.jsonl
timeout 600 python tools/cu_partition_probe.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|model = MAGNET" > $O/cu_partition_probe.jsonl   # step halves side by side on disjoint CU partitions
PROBE_PACKS=1 timeout 600 python tools/cu_partition_probe.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|model = MAGNET" > $O/cu_partition_probe_packs.jsonl   # only the packs on the small partition
timeout 300 python tools/power_probe.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|model = MAGNET" > $O/power_probe.jsonl   # socket power / shader clock per part of the step (rocm-smi)
for u in issue_rate gather_rate; do
  [ -x tools/ubench/$u ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ubench/$u tools/ubench/$u.hip 2>/dev/null
  timeout 120 tools/ubench/$u > $O/$u.txt 2>&1
done
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
for f in $(find $O/overlap_trace -name "*kernel_trace.csv"); do python - "$f" "$O/overlap_kernel_trace_tail.txt" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
keep = [r for r in rows if any(k in r["Kernel_Name"] for k in ("conv_mfma", "cv_v3", "pack_", "upsample"))][-40:]
with open(sys.argv[2], "w") as f:
    for r in keep:
        f.write(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:12.1f} us  +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:9.1f} us  q{r.get("Queue_Id", "?")}  {r["Kernel_Name"][:90]}\n')
PY
done
for f in $(find $O -name "*kernel_trace.csv"); do head -200 $f > $f.head; rm $f; done
tail -1 $O/bench.json | cut -c1-600
