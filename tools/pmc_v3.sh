#!/bin/bash
# Dev tool (GPU box): SQ / TA counters of the matcher kernel alone (bench.py --kernel-only), three short passes.
# usage: tools/pmc_v3.sh <tag> <path incl. dev bits>
tag=$1; path=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag; mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 60 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p -- python bench.py --kernel-only --path $path --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python - <<PY
import csv,glob,collections
agg=collections.defaultdict(lambda: collections.defaultdict(list))
dur=collections.defaultdict(list)
for f in glob.glob("$out/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"]
        if "cv_" not in k: continue
        agg[k.split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k.split("(")[0][:90]].append(float(r["End_Timestamp"])-float(r["Start_Timestamp"]))
for k,d in agg.items():
    print("TAG $tag PATH $path KERNEL",k, "mean dispatch ns (profiled)", sum(dur[k])/len(dur[k]))
    for c,v in sorted(d.items()): print(f"  {c:40s} mean {sum(v)/len(v):18.1f}  n={len(v)}")
PY
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete; rm -rf $out/p*/
