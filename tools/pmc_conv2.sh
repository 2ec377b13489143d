#!/bin/bash
# Dev tool (GPU box): counters of the convolution kernels inside the default bench step (safe counter sets, short timeouts).
# usage: tools/pmc_conv2.sh <tag> [env assignments / bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcconv_$tag; mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" \
            "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 60 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustain-s 0 "$@" > $out/p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_mfma_kernel" not in k: continue
        key=k.split("(")[0][:80]+" grid="+r["Grid_Size"]
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,d in agg.items():
    print("TAG $tag KERNEL",k, "mean dispatch ns (profiled)", sum(dur[k])/len(dur[k]))
    for c,v in sorted(d.items()): print(f"  {c:40s} mean {sum(v)/len(v):18.1f}  n={len(v)}")
PY
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete
