#!/bin/bash
# Second round on the N > 1 path's burst (ZKE_BENCH_FORCE_DIST=1, one rank; communicator after the engine, device-wide wait only):
# slots warmed again after the communicator's first use (REWARM), the region's barriers on a gloo group (GLOO_BARRIER).
set -o pipefail
out=gpurun_out/disttail2; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
export ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 ZKE_BENCH_X_LATE_PG=1 ZKE_BENCH_X_NO_ENG_SYNC=1
: > $out/summary.txt
for cfg in "rewarm ZKE_BENCH_X_REWARM=1" "rewarm_gloo ZKE_BENCH_X_REWARM=1 ZKE_BENCH_X_GLOO_BARRIER=1" "gloo ZKE_BENCH_X_GLOO_BARRIER=1"; do
  set -- $cfg; name=$1; shift
  for st in "20 5 a" "20 5 b" "2000 100 s"; do read steps warm tag <<< "$st"
    env "$@" timeout -k 10 300 python bench.py $B --steps $steps --warmup $warm > $out/${name}_$tag.json 2> $out/${name}_$tag.err || { tail -20 $out/${name}_$tag.err; exit 1; }
    echo "$name steps $steps: $(grep -o '"value": [0-9.]*' $out/${name}_$tag.json) $(grep tail_times $out/${name}_$tag.err)" | tee -a $out/summary.txt
  done
done
