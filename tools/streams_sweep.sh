#!/bin/bash
# throughput against batches in flight (and HW queues): bash tools/streams_sweep.sh "8 12 16 20 24 32"
for s in ${1:-8 12 16 20 24 32}; do
  v=$(python bench.py --steps 1500 --warmup 80 --no-cpu --no-saturated --streams $s 2>/dev/null | grep -o '"value": [0-9.]*'); echo "streams=$s $v"
done
for q in 24 32; do
  v=$(GPU_MAX_HW_QUEUES=$q python bench.py --steps 1500 --warmup 80 --no-cpu --no-saturated --streams $q 2>/dev/null | grep -o '"value": [0-9.]*'); echo "streams=$q queues=$q $v"
done
