#!/bin/bash
# A/B two engine builds on the same box: bash tools/ab.sh ab/head.so ab/new.so [reps]
A=$1; B=$2; N=${3:-3}
for r in $(seq $N); do for L in $A $B; do
  v=$(ZKE_LIB=$PWD/$L python bench.py --steps 1500 --warmup 80 --no-cpu 2>/dev/null | grep -o '"value": [0-9.]*')
  echo "$L $v"
done; done
