O=gpurun_out/r4g; mkdir -p $O
ABLATE_V3=1 timeout 300 python tools/ablate.py C2 64 split 2>/dev/null | tee $O/ablate_v3_variants.log
