#!/bin/bash
# Dev tool (GPU box): counters of the 3x3 + tail kernels in tools/bench_conv_mx.py (bf16x3 vs fp16 + e4m3 operand format).
tag=${1:-mx}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcconv_$tag; mkdir -p $out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE TA_BUSY_avr"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p -- python tools/bench_conv_mx.py 64 > $out/p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_mfma_kernel" not in k: continue
        key=k.split("(")[0][:90]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,d in sorted(agg.items()):
    print("KERNEL",k, "mean dispatch ns (profiled)", round(sum(dur[k])/len(dur[k])), "n", len(dur[k]))
    for c,v in sorted(d.items()): print(f"  {c:40s} mean {sum(v)/len(v):18.1f}")
PY
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete; rm -rf $out/p*/
