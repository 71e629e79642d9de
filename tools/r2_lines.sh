#!/bin/bash
out=gpurun_out/r2p; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_verify.py tests/test_gpu_bench_contract.py tests/test_gpu_cpp_mirror.py tests/test_rfc8463_vector.py -m gpu -x -q > $out/tests_quick.log 2>&1 || { tail -30 $out/tests_quick.log; exit 1; }
tail -2 $out/tests_quick.log
timeout -k 10 400 python bench.py > $out/bench_line.json 2> $out/bench_line.err || { tail -20 $out/bench_line.err; exit 1; }
for k in 1 2 3; do timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_line$k.json 2> $out/bench_driver_line.err || { tail -20 $out/bench_driver_line.err; exit 1; }; done
python -c "
import json
for f in ('bench_line','bench_driver_line1','bench_driver_line2','bench_driver_line3'):
    j=json.load(open('$out/'+f+'.json')); print(f, j['value'], j['ms_per_step'], j['host_submit_ms'], j['roofline']['achieved'], j['roofline']['launch_us'], j['roofline']['alone'], j['roofline']['aggregate']['achieved'], j['cpu_baseline']['value'], j['sha256_saturated']['achieved_GBps'])"
