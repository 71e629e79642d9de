#!/bin/bash
# Every bench workload once (GPU box): bash tools/workloads_sweep.sh -> gpurun_out/workloads.txt
set -o pipefail
mkdir -p gpurun_out
: > gpurun_out/workloads.txt
for W in c2 c3 c5 c5re c2ed c4shard; do
  S=1000; [ $W = c4shard ] && S=200; [ $W = c2ed ] && S=300
  timeout -k 10 400 python bench.py --workload $W --steps $S --warmup 40 --no-cpu --no-saturated 2>gpurun_out/wl_$W.err | grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*\|"kernels_us": {[^}]*}' | tr '\n' ' ' | sed "s|^|$W |" >> gpurun_out/workloads.txt || { tail -5 gpurun_out/wl_$W.err; exit 1; }
  echo >> gpurun_out/workloads.txt
done
cat gpurun_out/workloads.txt
