#!/bin/bash
# Where the end of the timed region goes with the N > 1 code path on (ZKE_BENCH_FORCE_DIST=1, one rank), and two experiments:
# the communicator created after the engine's slots (ZKE_BENCH_X_LATE_PG), the device-wide wait alone (ZKE_BENCH_X_NO_ENG_SYNC).
# -> gpurun_out/disttail/
set -o pipefail
out=gpurun_out/disttail; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
export ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1
: > $out/summary.txt
for cfg in "base ZKE_X=0" "late ZKE_BENCH_X_LATE_PG=1" "nosync ZKE_BENCH_X_NO_ENG_SYNC=1" "late_nosync ZKE_BENCH_X_LATE_PG=1 ZKE_BENCH_X_NO_ENG_SYNC=1"; do
  set -- $cfg; name=$1; shift
  for st in "20 5 a" "20 5 b" "2000 100 s"; do read steps warm tag <<< "$st"
    env "$@" timeout -k 10 300 python bench.py $B --steps $steps --warmup $warm > $out/${name}_$tag.json 2> $out/${name}_$tag.err || { tail -20 $out/${name}_$tag.err; exit 1; }
    echo "$name steps $steps: $(grep -o '"value": [0-9.]*' $out/${name}_$tag.json) $(grep tail_times $out/${name}_$tag.err)" | tee -a $out/summary.txt
  done
done
