#!/bin/bash
# A/B: the front end compiled for 6 (default) / 5 / 4 waves per SIMD on the tree with the lane-per-tag parser (variants/libzke_p5.so, p4)
out=gpurun_out/r2ab15; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 4"
for rep in 1 2; do for v in default p5 p4; do
  if [ $v = default ]; then unset ZKE_LIB; else export ZKE_LIB=$PWD/variants/libzke_$v.so; fi
  for st in "2000 100" "20 5"; do set -- $st
    timeout -k 10 200 python bench.py $B --steps $1 --warmup $2 2>$out/$v.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v steps $1:', round(j['value']/1e6,2), 'M e-mails/s', j['ms_per_step'], 'ms/step, front end alone', j['kernels_us_alone']['parse_us'])" || { tail -3 $out/$v.err; }
  done; done; done
