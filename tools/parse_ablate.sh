#!/bin/bash
# Front-end latency ablation: parse_kernel truncated after stage k (ZKE_DEBUG_PARSE_STOP=k), one batch at a time.
# Usage (GPU box): bash tools/parse_ablate.sh > gpurun_out/parse_ablate.txt
for k in 1 2 3 4 5 6 7 0; do
  v=$(ZKE_DEBUG_PARSE_STOP=$k ZKE_BENCH_NOCHECK=1 python bench.py --steps 100 --warmup 10 --no-cpu --streams 1 2>/dev/null | grep -o 'parse_us[^,]*')
  echo "stop=$k $v"
done
