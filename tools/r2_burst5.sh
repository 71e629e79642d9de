#!/bin/bash
# the driver's command, as the driver runs it (CPU baseline and all), five times in fresh processes
out=gpurun_out/r2b5; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q 2>&1 | tail -2
for k in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > $out/b$k.json 2>$out/b$k.err || { tail -5 $out/b$k.err; exit 1; }
  python -c "import json; j=json.load(open('$out/b$k.json')); print('run $k', round(j['value']/1e6,2), 'M e-mails/s', j['ms_per_step'], 'ms/step  cpu', round(j['cpu_baseline']['value']))"
done
