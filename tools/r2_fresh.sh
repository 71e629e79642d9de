#!/bin/bash
# A fresh box's first runs of the driver's command (every slot primed with a real batch before the warm-up), the N > 1 rehearsal, the contract tests.
set -o pipefail
out=gpurun_out/fresh; mkdir -p $out
export TMPDIR=/tmp
: > $out/summary.txt
for i in 1 2 3; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/b$i.json 2> $out/b$i.err || { tail -20 $out/b$i.err; exit 1; }
  echo "run $i: $(grep -o '"value": [0-9.]*' $out/b$i.json | head -1) $(grep -o '"host_submit_ms": [0-9.]*' $out/b$i.json) $(grep -o '"warmup_effective": [0-9]*' $out/b$i.json)" | tee -a $out/summary.txt
done
ZKE_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-saturated > $out/dist.json 2> $out/dist.err || { tail -20 $out/dist.err; exit 1; }
echo "dist: $(grep -o '"value": [0-9.]*' $out/dist.json | head -1)" | tee -a $out/summary.txt
timeout -k 10 500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -1 $out/tests.log | tee -a $out/summary.txt
