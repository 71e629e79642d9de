#!/bin/bash
# The driver's command five times in a row on one box (default queue cap), then twice under the old roomy pool (26).
set -o pipefail
out=gpurun_out/burst5; mkdir -p $out
export TMPDIR=/tmp
: > $out/summary.txt
for i in 1 2 3 4 5; do
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/b$i.json 2> $out/b$i.err || { tail -20 $out/b$i.err; exit 1; }
  echo "cap 23 run $i: $(grep -o '"value": [0-9.]*' $out/b$i.json | head -1)" | tee -a $out/summary.txt
done
for i in 1 2; do
  GPU_MAX_HW_QUEUES=26 timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu > $out/q26_$i.json 2> $out/q26_$i.err || { tail -20 $out/q26_$i.err; exit 1; }
  echo "cap 26 run $i: $(grep -o '"value": [0-9.]*' $out/q26_$i.json | head -1)" | tee -a $out/summary.txt
done
