#!/bin/bash
# Dev tool (GPU box): PMC passes for the conv kernel inside the default bench step.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_conv; mkdir -p $out; i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/p$i.log 2>&1
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_mfma_kernel<8" not in k: continue
        # split 3x3 vs 1x1 by grid? use LDS/duration unknown -> bucket by counter magnitude later
        agg[r["Counter_Name"]][r["Dispatch_Id"]]=float(r["Counter_Value"])
import statistics
for c,d in sorted(agg.items()):
    v=sorted(d.values())
    big=[x for x in v if x>=0.5*max(v)] if max(v)>0 else v
    print(f"{c:30s} all-mean {sum(v)/len(v):16.1f} big-launch-mean {sum(big)/len(big):16.1f} n={len(v)}/{len(big)}")
PY
