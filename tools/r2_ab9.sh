#!/bin/bash
# front-end register budget: 6 / 5 / 4 waves per SIMD (80 / 96 / 120 VGPRs; scratch 368 / 360 / 0 bytes per lane)
out=gpurun_out/ab9; mkdir -p $out
for v in base pw5 pw4 base pw4; do
  L=""; [ $v != base ] && L="variants/libzke_$v.so"
  ZKE_LIB=$L python bench.py --no-cpu --no-saturated --steps 2000 --warmup 100 2>$out/$v.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', j['value'], j['ms_per_step'], j['kernels_us_alone']['parse_us'])"
done
