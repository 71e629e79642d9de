#!/bin/bash
# The N > 1 code path on one GPU (ZKE_BENCH_FORCE_DIST=1: RCCL init, exchange, barrier with one rank) against the plain
# one: does the communicator's stream cost a slot its hardware queue?  GPU_MAX_HW_QUEUES swept around the default (S + 4).
# -> gpurun_out/distq/summary.txt
set -o pipefail
out=gpurun_out/distq; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
: > $out/summary.txt
run() {  # label, env..., -- args
  label=$1; shift
  envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 10 300 python bench.py $B "$@" > $out/$label.json 2> $out/$label.err || { tail -20 $out/$label.err; exit 1; }
  python -c "
import json; j=json.load(open('$out/$label.json')); print('$label', round(j['value']/1e6,2), 'M e-mails/s', j['ms_per_step'], 'ms/step')" | tee -a $out/summary.txt
}
for rep in 1 2; do
  run plain_burst_$rep ZKE_X=0 -- --steps 20 --warmup 5
  run dist_burst_$rep ZKE_BENCH_FORCE_DIST=1 -- --steps 20 --warmup 5
done
run plain_steady ZKE_X=0 -- --steps 2000 --warmup 100
run dist_steady ZKE_BENCH_FORCE_DIST=1 -- --steps 2000 --warmup 100
for q in 28 30 32; do
  run dist_steady_q$q ZKE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=$q -- --steps 2000 --warmup 100
  run dist_burst_q$q ZKE_BENCH_FORCE_DIST=1 GPU_MAX_HW_QUEUES=$q -- --steps 20 --warmup 5
done
cat $out/summary.txt
