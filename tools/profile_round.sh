#!/bin/bash
# GPU box: official bench line + rocprofv3 kernel stats + HBM traffic counters for the round.
# usage: tools/profile_round.sh <round-tag>
tag=${1:-r1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
python bench.py > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag/stats -o s -- python bench.py --no-cpu-baseline > gpurun_out/$tag/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/fetch -o f -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$tag/write -o w -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$tag/write.log 2>&1
# calibration of FETCH_SIZE on a known streaming read (a 1 GiB torch copy)
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$tag/calib -o c -- python -c "
import torch
x=torch.empty(1<<28,dtype=torch.float32,device='cuda'); y=torch.empty_like(x)
for _ in range(3): y.copy_(x)
torch.cuda.synchronize()" > gpurun_out/$tag/calib.log 2>&1
python - <<PY
import csv,glob,collections,json
def agg(pattern, key):
    d=collections.defaultdict(list)
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0][:70]].append(float(r[key]))
    return d
for name,pat in (("FETCH_SIZE","gpurun_out/$tag/fetch/**/*counter_collection.csv"),("WRITE_SIZE","gpurun_out/$tag/write/**/*counter_collection.csv"),("CALIB FETCH_SIZE","gpurun_out/$tag/calib/**/*counter_collection.csv")):
    d=agg(pat,"Counter_Value")
    print("==",name,"(KB per dispatch, mean)")
    for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:8]:
        print(f"  {k:70s} {sum(v)/len(v):14.1f}  n={len(v)}")
PY
head -40 $(find gpurun_out/$tag/stats -name "*kernel_stats.csv" | head -1) | cut -c1-220
tail -1 gpurun_out/$tag/bench.json
find gpurun_out/$tag -name "*.db" -delete; find gpurun_out/$tag -name "*_agent_info.csv" -delete
# keep traces small: drop the raw kernel trace of the stats run beyond the summary
for f in $(find gpurun_out/$tag -name "*kernel_trace.csv"); do head -400 $f > $f.head; rm $f; done
