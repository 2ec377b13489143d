#!/bin/bash
# Dev tool (GPU box): PMC passes for the convolution kernels inside one F-Net forward (tools/bench_fnet.py).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_fnet; mkdir -p $out; i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU" \
            "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p -- python tools/bench_fnet.py --frames 8 --steps 1 --skip-torch > $out/p$i.log 2>&1
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "conv_mfma_kernel" not in k: continue
        name=k.split("conv_mfma_kernel")[1].split("(")[0]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name,d in sorted(agg.items()):
    print("== conv_mfma_kernel"+name)
    for c,v in sorted(d.items()):
        v=sorted(v); big=[x for x in v if x>=0.5*max(v)] if max(v)>0 else v
        print(f"   {c:30s} mean {sum(v)/len(v):16.1f}  big-launch-mean {sum(big)/len(big):16.1f}  n={len(v)}/{len(big)}")
PY
