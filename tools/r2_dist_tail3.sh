#!/bin/bash
# Third round on the N > 1 path (ZKE_BENCH_FORCE_DIST=1, one rank): HSA_NO_SCRATCH_RECLAIM=1 (ROCr keeps a queue's scratch memory
# instead of taking it back when another queue — the communicator's kernels use a lot — asks for more than the pool holds).
set -o pipefail
out=gpurun_out/disttail3; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
export ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1
: > $out/summary.txt
for cfg in "nsr HSA_NO_SCRATCH_RECLAIM=1" "nsr_nosync HSA_NO_SCRATCH_RECLAIM=1 ZKE_BENCH_X_NO_ENG_SYNC=1" "nsr_late_nosync HSA_NO_SCRATCH_RECLAIM=1 ZKE_BENCH_X_LATE_PG=1 ZKE_BENCH_X_NO_ENG_SYNC=1"; do
  set -- $cfg; name=$1; shift
  for st in "20 5 a" "20 5 b" "2000 100 s"; do read steps warm tag <<< "$st"
    env "$@" timeout -k 10 300 python bench.py $B --steps $steps --warmup $warm > $out/${name}_$tag.json 2> $out/${name}_$tag.err || { tail -20 $out/${name}_$tag.err; exit 1; }
    echo "$name steps $steps: $(grep -o '"value": [0-9.]*' $out/${name}_$tag.json) $(grep tail_times $out/${name}_$tag.err)" | tee -a $out/summary.txt
  done
done
unset ZKE_BENCH_FORCE_DIST
for st in "20 5 a" "2000 100 s"; do read steps warm tag <<< "$st"
  HSA_NO_SCRATCH_RECLAIM=1 timeout -k 10 300 python bench.py $B --steps $steps --warmup $warm > $out/plain_nsr_$tag.json 2> $out/plain_nsr_$tag.err || { tail -20 $out/plain_nsr_$tag.err; exit 1; }
  echo "plain_nsr steps $steps: $(grep -o '"value": [0-9.]*' $out/plain_nsr_$tag.json)" | tee -a $out/summary.txt
done
