O=gpurun_out/r5; mkdir -p $O
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_torchrun_n1.json 2> $O/bench_torchrun_n1.err
echo "rc=$? lines=$(wc -l < $O/bench_torchrun_n1.json)"; cut -c1-300 $O/bench_torchrun_n1.json; python -c "
import json;d=json.load(open('$O/bench_torchrun_n1.json'));print(d['value'], d['rccl'], d['roofline']['frac'], d['roofline']['bound'])"
python bench.py --gpus 1 > $O/bench_plain.json 2> $O/bench_plain.err; echo "rc=$? lines=$(wc -l < $O/bench_plain.json)"
python -c "
import json;d=json.load(open('$O/bench_plain.json'));print(d['value'], d['rccl'], d['roofline']['frac'], d['roofline_conv'].get('mfma_busy'))"
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_fast_matcher.py -m gpu -q 2>&1 | tail -n 1; done
