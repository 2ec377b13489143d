#!/bin/bash
# GPU session r2a: first measurement of the production matcher (tests, A/B timing, counters for both matchers)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py -q -x -m gpu > $O/pytest_fast.log 2>&1; tail -15 $O/pytest_fast.log
for wl in C2 C1 C4 C5 shipped C2L; do
  fr=64; [ $wl = C4 ] && fr=24; [ $wl = C5 ] && fr=30; [ $wl = shipped ] && fr=47; [ $wl = C2L ] && fr=4
  timeout 200 python tools/ablate.py $wl $fr >> $O/ablate.log 2>&1
done
timeout 200 python tools/ablate.py C2 64 split >> $O/ablate.log 2>&1
cat $O/ablate.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; tail -1 $O/bench.json | cut -c1-600
for path in 0 2; do
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SMEM" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
            "SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
            "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmc_path$path/p$i -o p -- python bench.py --kernel-only --path $path --steps 3 --warmup 1 --no-cpu-baseline > $O/pmc_path${path}_p$i.log 2>&1
done
done
python - <<PY
import csv,glob,collections
for path in (0,2):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"$O/pmc_path{path}/p*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"]
            if "cv_" not in k: continue
            agg[k.split("(")[0][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,d in agg.items():
        print("PATH",path,"KERNEL",k)
        for c,v in sorted(d.items()): print(f"  {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
PY
find $O -name "*.db" -delete; find $O -name "*_agent_info.csv" -delete
