#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed profiles/<tag>/ records:
bench line, rocprofv3 kernel stats, the counter rows of our kernels, traffic_C2.json (FETCH_SIZE / WRITE_SIZE corrected by the
calibration copy of the same session) with the matcher's SQ counters, kernel-only lines of every config, extra runs."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r6"
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def rows_of(pat):
    for f in glob.glob(pat, recursive=True):
        yield from csv.DictReader(open(f))


# ---- counters per kernel (mean over dispatches) ----
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows_of(os.path.join(src, "pmc*", "**", "*counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0]
    if "magnet::" not in k:
        continue
    ctr[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
calib = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v = [float(r["Counter_Value"]) for r in rows_of(os.path.join(src, f"calib_{c}", "**", "*counter_collection.csv"))
         if "copy" in r["Kernel_Name"].lower() and r["Counter_Name"] == c]
    calib[c] = sum(v) / len(v) if v else None
GiB_KB = float(1 << 20)
fcorr = GiB_KB / calib["FETCH_SIZE"] if calib["FETCH_SIZE"] else 2.0
wcorr = GiB_KB / calib["WRITE_SIZE"] if calib["WRITE_SIZE"] else 1.0
with open(os.path.join(dst, "pmc_kernels.txt"), "w") as o:
    o.write(f"# mean per dispatch over the rocprofv3 --pmc passes of `python bench.py --steps 20 --warmup 5` (tools/profile_round.sh)\n"
            f"# calibration copy (1 GiB read + 1 GiB write): FETCH_SIZE {calib['FETCH_SIZE']} KB, WRITE_SIZE {calib['WRITE_SIZE']} KB"
            f" -> corrections x{fcorr:.3f}, x{wcorr:.3f}\n")
    for k, d in sorted(ctr.items()):
        o.write(f"KERNEL {k}   mean dispatch {sum(dur[k]) / len(dur[k]) / 1e3:.1f} us (profiled)\n")
        for c, v in sorted(d.items()):
            o.write(f"  {c:36s} {sum(v) / len(v):18.1f}  n={len(v)}\n")

mk = sorted((k for k in ctr if "cv_v3_kernel" in k or "cv_fast64_kernel" in k or "cv_fast_kernel" in k), key=lambda k: "cv_v3" not in k)
out = {"workload": "C2, 64 frames/launch, bf16 features, split-bf16 cost output (inside the bench step)",
       "calib_copy_1GiB_KB": calib, "fetch_correction": fcorr, "write_correction": wcorr}
if mk:
    k = mk[0]
    m = {c: sum(v) / len(v) for c, v in ctr[k].items()}
    fetch, write = m.get("FETCH_SIZE", 0) * 1024 * fcorr, m.get("WRITE_SIZE", 0) * 1024 * wcorr
    iters = 64 * 120 * 160 * 4                                     # (pixel, view) wave iterations per launch
    us = sum(dur[k]) / len(dur[k]) / 1e3
    out.update({"kernel": k.replace("void magnet::", ""), "traffic_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write,
                "algorithmic_bytes_per_launch": 1150156800,
                # GRBM_GUI_ACTIVE is summed over the 8 XCDs; TA_BUSY_avr is the mean over the texture-address units
                "ta_busy_frac": m.get("TA_BUSY_avr", 0) / (m["GRBM_GUI_ACTIVE"] / 8) if m.get("GRBM_GUI_ACTIVE") else None,
                "clock_ghz_during_kernel": m["GRBM_GUI_ACTIVE"] / 8 / (us * 1e3) if m.get("GRBM_GUI_ACTIVE") and us else None,
                "sq": {"valu_insts_per_pixel_view": m.get("SQ_INSTS_VALU", 0) / iters, "salu_insts_per_pixel_view": m.get("SQ_INSTS_SALU", 0) / iters,
                       "vmem_insts_per_pixel_view": m.get("SQ_INSTS_VMEM_RD", 0) / iters, "lds_insts_per_pixel_view": m.get("SQ_INSTS_LDS", 0) / iters,
                       "l1_accesses_per_pixel_view": m.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / iters,
                       "valu_busy_frac": m.get("SQ_ACTIVE_INST_VALU", 0) * 4 / (1024 * m["GRBM_GUI_ACTIVE"] / 8) if m.get("GRBM_GUI_ACTIVE") else None,
                       "wave_cycles_waiting_frac": m.get("SQ_WAIT_ANY", 0) / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
                       "l2_hit_frac": m.get("TCC_HIT_sum", 0) / max(1.0, m.get("TCC_HIT_sum", 0) + m.get("TCC_MISS_sum", 0)),
                       "profiled_launch_us": us,
                       "note": "valu_busy_frac = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs): busy share of the cycles the chip actually ran"}})
# the same kernel alone and warm (pmc_alone pass): last 100 dispatches
al = [r for r in rows_of(os.path.join(src, "pmc_alone", "**", "*counter_collection.csv")) if "cv_v3_kernel" in r["Kernel_Name"]]
if al:
    g = [r for r in al if r["Counter_Name"] == "GRBM_GUI_ACTIVE"][-100:]
    t = [r for r in al if r["Counter_Name"] == "TA_BUSY_avr"][-100:]
    if g:
        cyc = sum(float(r["Counter_Value"]) for r in g) / len(g) / 8
        ns = sum(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in g) / len(g)
        out["alone_warm"] = {"profiled_launch_us": ns / 1e3, "clock_ghz_during_kernel": cyc / ns,
                             "ta_busy_frac": (sum(float(r["Counter_Value"]) for r in t) / len(t) / cyc) if t else None}
json.dump(out, open(os.path.join(dst, "traffic_C2.json"), "w"), indent=1)

shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_C2.json"))
st = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(dst, "bench_C2_kernel_stats.csv"), "w") as o:
        w = csv.writer(o); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:16]:
            w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
for f in ("configs.jsonl", "ablate_C2_split.log", "ablate_tx.log", "ablate_halfq.log", "clock_recovery.jsonl", "cu_partition_probe.jsonl", "cu_partition_probe_packs.jsonl", "power_probe.jsonl", "overlap_ab.jsonl", "overlap_kernel_trace_tail.txt", "frames_sweep.jsonl", "wino_bound.jsonl", "issue_rate.txt", "gather_rate.txt", "bench_fnet.json",
          "fnet_layers.txt", "fvolume_bench.jsonl", "bench_end_to_end.json", "bench_pipeline.json", "kernel_only_C2_nchw.json", "parity_stats_gpu_tests.txt"):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f if f != "configs.jsonl" else "matcher_kernel_only_all_configs.jsonl"))
for sdir, oname in (("stats_shipped", "bench_shipped_kernel_stats.csv"), ("fvolume_stats", "fvolume_kernel_stats.csv")):
    st = glob.glob(os.path.join(src, sdir, "**", "*kernel_stats.csv"), recursive=True)
    if not st:
        continue
    rows = list(csv.DictReader(open(st[0])))
    with open(os.path.join(dst, oname), "w") as o:
        w = csv.writer(o); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows[:12]:
            w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
os.makedirs(os.path.join(dst, "extra"), exist_ok=True)
for f in glob.glob(os.path.join(src, "extra", "*.json")):
    shutil.copy(f, os.path.join(dst, "extra", os.path.basename(f)))
print(json.dumps(out, indent=1)[:3000])
