#!/bin/bash
# The N > 1 path (ZKE_BENCH_FORCE_DIST=1, one rank) against the slot count: is the communicator's handful of streams what pushes
# 22 slots over the number of hardware queues the chip schedules without time-slicing?  -> gpurun_out/distslots/summary.txt
set -o pipefail
out=gpurun_out/distslots; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
export ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 ZKE_BENCH_X_NO_ENG_SYNC=1
: > $out/summary.txt
for late in 1 0; do for S in 14 16 18 20; do
  for st in "20 5 a" "20 5 b" "2000 100 s"; do read steps warm tag <<< "$st"
    name=late${late}_S$S
    ZKE_BENCH_X_LATE_PG=$late timeout -k 10 300 python bench.py $B --streams $S --steps $steps --warmup $warm > $out/${name}_$tag.json 2> $out/${name}_$tag.err || { tail -20 $out/${name}_$tag.err; exit 1; }
    echo "$name steps $steps: $(grep -o '"value": [0-9.]*' $out/${name}_$tag.json) $(grep -o 'drained [0-9.]*' $out/${name}_$tag.err)" | tee -a $out/summary.txt
  done
done; done
