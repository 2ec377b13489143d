#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (written by tools/profile_round.sh on the GPU box) into the committed profiles/<tag>/ records:
bench line, rocprofv3 kernel stats, the PMC rows of our kernels and traffic_C2.json (FETCH_SIZE x calibration + WRITE_SIZE)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
src, dst = os.path.join(R, "gpurun_out", tag), os.path.join(R, "profiles", tag)
os.makedirs(dst, exist_ok=True)


def agg(pat):
    d = collections.defaultdict(list)
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return d


fe, wr, ca = (agg(os.path.join(src, s, "**", "*counter_collection.csv")) for s in ("fetch", "write", "calib"))
calib = [v for k, v in ca.items() if "copyBuffer" in k][0]
calib_kb = sum(calib) / len(calib)
corr = (1 << 20) / calib_kb                      # the calibration copy reads 1 GiB = 2^20 KiB


def kern(sub):
    f = [x for k, v in fe.items() if sub in k for x in v]
    w = [x for k, v in wr.items() if sub in k for x in v]
    return sum(f) / len(f), sum(w) / len(w), len(f)


out = {"workload": "C2, 64 frames/launch, bf16 features", "calib_copy_1GiB_FETCH_SIZE_KB": calib_kb, "fetch_correction": corr, "kernels": {}}
for name, sub, alg in (("cost_volume", "cv_cand_kernel", 1150156800), ("conv_gnet_stack_fused", "conv_mfma_kernel<8, 2, 128, 1, 1>", None),
                       ("conv_mask_stack_fused", "conv_mfma_kernel<8, 2, 128, 1, 9>", None)):
    f, w, n = kern(sub)
    out["kernels"][name] = {"kernel": sub, "FETCH_SIZE_KB_raw": f, "WRITE_SIZE_KB_raw": w, "fetch_bytes_corrected": f * 1024 * corr,
                            "write_bytes": w * 1024, "traffic_bytes_per_launch": f * 1024 * corr + w * 1024,
                            "algorithmic_bytes_per_launch": alg, "dispatches": n}
cv = out["kernels"]["cost_volume"]
out.update({"traffic_bytes_per_launch": cv["traffic_bytes_per_launch"], "algorithmic_bytes_per_launch": 1150156800,
            "kernel": [k for k in fe if "cv_cand_kernel" in k][0].split("(")[0] + " (split-bf16 channel-last cost output)",
            "note": "FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc passes (tools/profile_round.sh); FETCH_SIZE corrected by the "
                    "factor measured on a 1 GiB device copy in the same session (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md "
                    "§HBM; Infinity-Cache hits are counted too); WRITE_SIZE uncalibrated."})
json.dump(out, open(os.path.join(dst, "traffic_C2.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, "bench_C2.json"))
rows = list(csv.DictReader(open(glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)[0])))
with open(os.path.join(dst, "bench_C2_kernel_stats.csv"), "w") as o:
    w = csv.writer(o); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows[:16]:
        w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
for nm, s in (("pmc_FETCH_SIZE.csv", "fetch"), ("pmc_WRITE_SIZE.csv", "write"), ("pmc_calib_copy_FETCH_SIZE.csv", "calib")):
    f = glob.glob(os.path.join(src, s, "**", "*counter_collection.csv"), recursive=True)[0]
    with open(os.path.join(dst, nm), "w") as o:
        w = csv.writer(o); w.writerow(["Kernel_Name", "Counter_Name", "Counter_Value", "Dispatch_Id"])
        for r in csv.DictReader(open(f)):
            if any(t in r["Kernel_Name"] for t in ("magnet::", "copyBuffer")):
                w.writerow([r["Kernel_Name"][:120], r["Counter_Name"], r["Counter_Value"], r["Dispatch_Id"]])
print(json.dumps(cv, indent=1))
