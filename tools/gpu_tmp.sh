O=gpurun_out/r4o; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_rays_poses.py -x -q -m gpu -s -k "full_resolution or large_batch or oracle_forward or C5_end" > $O/new_parity.log 2>&1; echo "rc=$?" >> $O/new_parity.log
grep -v "^\[parity" $O/new_parity.log | tail -12; grep "full grid\|B = 4200" $O/new_parity.log | cut -c1-330
