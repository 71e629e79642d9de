#!/bin/bash
# The bench's final N > 1 flow on one GPU (forced through RCCL, one rank) beside the plain one, and the output-contract tests.
set -o pipefail
out=gpurun_out/distfinal; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -1 $out/tests.log
B="--no-cpu --no-saturated --alone-steps 0"
: > $out/summary.txt
for tag in a b c; do
  ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 timeout -k 10 300 python bench.py $B --gpus 1 --steps 20 --warmup 5 > $out/dist_$tag.json 2> $out/dist_$tag.err || { tail -20 $out/dist_$tag.err; exit 1; }
  echo "dist steps 20: lines on stdout $(wc -l < $out/dist_$tag.json) $(grep -o '"value": [0-9.]*' $out/dist_$tag.json) $(grep -o '"batches_in_flight": [0-9]*' $out/dist_$tag.json) $(grep tail_times $out/dist_$tag.err)" | tee -a $out/summary.txt
done
ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 timeout -k 10 300 python bench.py $B --steps 2000 --warmup 100 > $out/dist_s.json 2> $out/dist_s.err || { tail -20 $out/dist_s.err; exit 1; }
echo "dist steps 2000: $(grep -o '"value": [0-9.]*' $out/dist_s.json) $(grep tail_times $out/dist_s.err)" | tee -a $out/summary.txt
for tag in a b; do
  timeout -k 10 300 python bench.py $B --gpus 1 --steps 20 --warmup 5 > $out/plain_$tag.json 2> $out/plain_$tag.err || { tail -20 $out/plain_$tag.err; exit 1; }
  echo "plain steps 20: lines on stdout $(wc -l < $out/plain_$tag.json) $(grep -o '"value": [0-9.]*' $out/plain_$tag.json) $(grep -o '"batches_in_flight": [0-9]*' $out/plain_$tag.json)" | tee -a $out/summary.txt
done
