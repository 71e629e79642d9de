#!/bin/bash
# The N > 1 path on one GPU (forced through RCCL) with the hardware-queue pool CAPPED just above the slot count: the communicator's streams
# then share queues (they are idle while batches run) instead of adding queues past what the chip schedules without time-slicing.
set -o pipefail
out=gpurun_out/distcap; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
export ZKE_BENCH_FORCE_DIST=1
: > $out/summary.txt
for cfg in "22 23" "22 24" "22 25" "20 22" "20 23" "18 20"; do set -- $cfg; S=$1; Q=$2
  for st in "20 5 a" "20 5 b" "2000 100 s"; do read steps warm tag <<< "$st"
    name=S${S}_Q${Q}
    GPU_MAX_HW_QUEUES=$Q timeout -k 10 300 python bench.py $B --streams $S --steps $steps --warmup $warm > $out/${name}_$tag.json 2> $out/${name}_$tag.err || { tail -20 $out/${name}_$tag.err; exit 1; }
    echo "$name steps $steps: $(grep -o '"value": [0-9.]*' $out/${name}_$tag.json)" | tee -a $out/summary.txt
  done
done
