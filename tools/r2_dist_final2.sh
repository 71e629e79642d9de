#!/bin/bash
# Final defaults: the N > 1 flow on one GPU (18 slots under a queue cap of 20), the plain path with its roomy pool (26) and with a tight cap (23).
set -o pipefail
out=gpurun_out/distfinal2; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
: > $out/summary.txt
run() { name=$1; shift; envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py $B "$@" > $out/$name.json 2> $out/$name.err || { tail -20 $out/$name.err; exit 1; }
  echo "$name: $(grep -o '"value": [0-9.]*' $out/$name.json) $(grep -o '"batches_in_flight": [0-9]*' $out/$name.json) $(grep tail_times $out/$name.err)" | tee -a $out/summary.txt; }
for t in a b c; do run dist_burst_$t ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 -- --gpus 1 --steps 20 --warmup 5; done
run dist_steady ZKE_BENCH_FORCE_DIST=1 -- --steps 2000 --warmup 100
for t in a b c; do run plain_q23_burst_$t GPU_MAX_HW_QUEUES=23 -- --gpus 1 --steps 20 --warmup 5; done
run plain_q23_steady GPU_MAX_HW_QUEUES=23 -- --steps 2000 --warmup 100
for t in a b; do run plain_burst_$t ZKE_X=0 -- --gpus 1 --steps 20 --warmup 5; done
run plain_steady ZKE_X=0 -- --steps 2000 --warmup 100
timeout -k 10 500 python -m pytest tests/test_gpu_bench_contract.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -1 $out/tests.log
