#!/bin/bash
# hash / modexp launch register budget: 2 / 3 / 4 waves per SIMD (187 / 168 / 128 VGPRs; scratch 0 / 52 / 672 bytes per lane)
out=gpurun_out/ab10; mkdir -p $out
for v in base sw3 sw4 base sw3; do
  L=""; [ $v != base ] && L="variants/libzke_$v.so"
  ZKE_LIB=$L python bench.py --no-cpu --no-saturated --steps 2000 --warmup 100 2>$out/$v.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', j['value'], j['ms_per_step'], j['kernels_us_alone']['sha_us'], j['kernels_us_in_flight']['sha_us'])"
done
