#!/bin/bash
# Dev tool (GPU box): memory-pipeline counters (TA / TCP / TCC / TD / SQ) of the kernel-only bench for one `path` value.
# usage: tools/pmc_tc.sh <tag> <path> [bench args...]
tag=$1; path=$2; shift; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_$tag; mkdir -p $out
i=0
for ctrs in "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum" \
            "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" \
            "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_TA_TCP_STATE_READ_sum" \
            "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" "TCC_TAG_STALL_sum TCC_BUSY_avr TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
            "TD_TD_BUSY_sum TD_TC_STALL_sum TD_LOAD_WAVEFRONT_sum TD_SPI_STALL_sum" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
            "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/p$i -o p -- python bench.py --kernel-only --path $path --steps 3 --warmup 1 --no-cpu-baseline "$@" > $out/p$i.log 2>&1
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
find $out -name "*.db" -delete; find $out -name "*_agent_info.csv" -delete
