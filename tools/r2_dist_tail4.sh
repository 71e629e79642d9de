#!/bin/bash
# Fourth round on the N > 1 path's burst (ZKE_BENCH_FORCE_DIST=1, one rank; communicator after the engine, device-wide wait only).
set -o pipefail
out=gpurun_out/disttail4; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
export ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1 ZKE_BENCH_X_NO_ENG_SYNC=1
: > $out/summary.txt
for cfg in "A ZKE_BENCH_X_LATE_PG=1" "B_sleep ZKE_BENCH_X_LATE_PG=1 ZKE_BENCH_X_SLEEP=0.3" \
           "C_nowatchdog ZKE_BENCH_X_LATE_PG=1 TORCH_NCCL_ASYNC_ERROR_HANDLING=0 TORCH_NCCL_ENABLE_MONITORING=0" \
           "D_nopre_gloo ZKE_BENCH_X_LATE_PG=1 ZKE_BENCH_X_NO_PRE=1 ZKE_BENCH_X_GLOO_BARRIER=1" \
           "E_nopre_gloo_early ZKE_BENCH_X_NO_PRE=1 ZKE_BENCH_X_GLOO_BARRIER=1"; do
  set -- $cfg; name=$1; shift
  for tag in a b c; do
    env "$@" timeout -k 10 300 python bench.py $B --steps 20 --warmup 5 > $out/${name}_$tag.json 2> $out/${name}_$tag.err || { tail -20 $out/${name}_$tag.err; exit 1; }
    echo "$name: $(grep -o '"value": [0-9.]*' $out/${name}_$tag.json) $(grep tail_times $out/${name}_$tag.err)" | tee -a $out/summary.txt
  done
done
