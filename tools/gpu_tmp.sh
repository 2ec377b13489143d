#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/persist; mkdir -p $O
: > $O/abl_clock.log
for a in dev 1 7 8; do
  if [ $a = dev ]; then L=magnet_amd/libmagnet_hip_dev.so; else L=magnet_amd/libmagnet_hip_abl$a.so; fi
  rm -rf $O/pc
  CONV_LIB=$L MAGNET_CONV_VARIANT=8192 timeout 120 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pc -o p -- python tools/conv_kscale.py > /dev/null 2>&1
  python - $a $O/pc >> $O/abl_clock.log <<'PY'
import csv,glob,sys,collections
a,d=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_mfma_kernel" not in r["Kernel_Name"]: continue
        dur=float(r["End_Timestamp"])-float(r["Start_Timestamp"])
        key=round(dur/1e5)  # bucket by ~0.1 ms: the four input widths
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"])); agg[key]["dur"].append(dur)
for k in sorted(agg):
    v=agg[k]; dur=sum(v["dur"])/len(v["dur"]); g=sum(v["GRBM_GUI_ACTIVE"])/len(v["GRBM_GUI_ACTIVE"])/8; m=sum(v["SQ_VALU_MFMA_BUSY_CYCLES"])/len(v["SQ_VALU_MFMA_BUSY_CYCLES"])/1024
    print(f"lib {a:>3}: launch {dur/1e6:.3f} ms  clock {g/dur:.3f} GHz  matrix pipe busy {100*m/g:.1f} %")
PY
done
cat $O/abl_clock.log
