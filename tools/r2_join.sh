#!/bin/bash
# zke_engine_join: GPU parity of the slot / stream cases, then the N > 1 flow rehearsed on one GPU with the exchange enqueued behind a join.
set -o pipefail
out=gpurun_out/join; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_verify.py tests/test_gpu_bench_contract.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -40 $out/tests.log; exit 1; }
tail -1 $out/tests.log
B="--no-cpu --no-saturated --alone-steps 0"
: > $out/summary.txt
for tag in a b c d; do
  ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 timeout -k 10 300 python bench.py $B --gpus 1 --steps 20 --warmup 5 > $out/dist_$tag.json 2> $out/dist_$tag.err || { tail -20 $out/dist_$tag.err; exit 1; }
  echo "dist steps 20: $(grep -o '"value": [0-9.]*' $out/dist_$tag.json) $(grep tail_times $out/dist_$tag.err)" | tee -a $out/summary.txt
done
ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 timeout -k 10 300 python bench.py $B --steps 2000 --warmup 100 > $out/dist_s.json 2> $out/dist_s.err || { tail -20 $out/dist_s.err; exit 1; }
echo "dist steps 2000: $(grep -o '"value": [0-9.]*' $out/dist_s.json) $(grep tail_times $out/dist_s.err)" | tee -a $out/summary.txt
for tag in a b; do
  timeout -k 10 300 python bench.py $B --gpus 1 --steps 20 --warmup 5 > $out/plain_$tag.json 2> $out/plain_$tag.err || { tail -20 $out/plain_$tag.err; exit 1; }
  echo "plain steps 20: $(grep -o '"value": [0-9.]*' $out/plain_$tag.json)" | tee -a $out/summary.txt
done
