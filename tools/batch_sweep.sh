#!/bin/bash
# e-mails/s against e-mails per launch (dispatch-bound vs resource-bound): bash tools/batch_sweep.sh
for b in 256 512 1024 2048 4096 8192; do
  st=$(( 1500 * 1024 / b )); [ $st -lt 100 ] && st=100
  v=$(python bench.py --batch $b --steps $st --warmup 40 --no-cpu --no-saturated 2>/dev/null | grep -o '"value": [0-9.]*'); echo "batch=$b $v"
done
