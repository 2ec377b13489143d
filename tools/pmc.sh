#!/bin/bash
# Dev tool (GPU box): PMC passes over the kernel-only bench. usage: tools/pmc.sh <tag> [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag; mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TA_TA_BUSY_sum" \
            "GRBM_GUI_ACTIVE GRBM_COUNT" \
            "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p -- python bench.py --kernel-only --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p$i.log 2>&1
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "cv_" not in k: continue
        agg[k.split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,d in agg.items():
    print("KERNEL",k)
    for c,v in sorted(d.items()): print(f"  {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
