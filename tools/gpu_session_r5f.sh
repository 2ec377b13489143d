#!/bin/bash
O=gpurun_out/r5f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fast_matcher.py tests/test_gpu_conv.py -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --no-cpu-baseline > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err ) 2>&1 | grep real
python -c "
import json;d=json.load(open('$O/bench_torchrun_n1.json'));print(d['value'], d['rccl'], d['roofline']['frac'], d['roofline']['bound'], d['roofline_conv'].get('mfma_busy'))"
