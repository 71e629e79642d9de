#!/bin/bash
# A/B one engine build with and without an environment knob: bash tools/ab_env.sh ZKE_DEBUG_SKIP_ED=1 [reps]
K=$1; N=${2:-3}
for r in $(seq $N); do
  v=$(python bench.py --steps 1500 --warmup 80 --no-cpu 2>/dev/null | grep -o '"value": [0-9.]*'); echo "base   $v"
  v=$(env $K ZKE_BENCH_NOCHECK=1 python bench.py --steps 1500 --warmup 80 --no-cpu 2>/dev/null | grep -o '"value": [0-9.]*'); echo "$K $v"
done
