#!/bin/bash
# A/B of engine builds on the bench workload:  bash tools/ab_variants.sh NAME ...   (variants/libzke_NAME.so from tools/build_variant.sh;
# "base" = the in-tree library).  Steady state (2000 steps) and the driver's burst (20 steps), twice each, interleaved.
names=("$@")
for rep in 1 2; do for L in "${names[@]}"; do
  lib=variants/libzke_$L.so; [ $L = base ] && lib=zkemail.rs_amd/libzkemail_amd.so
  for S in "2000:100" "20:5"; do st=${S%%:*}; wu=${S##*:}
    ZKE_LIB=$lib python bench.py --steps $st --warmup $wu --no-cpu --no-saturated --no-e2e ${ZKE_AB_ARGS} 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$L', '$st steps', round(j['value']/1e6,2), 'M/s', j['ms_per_step'], 'alone', j['kernels_us_alone'] and {k:v for k,v in j['kernels_us_alone'].items() if v})"
  done
done; done
