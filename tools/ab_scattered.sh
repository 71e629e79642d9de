#!/bin/bash
# A/B of engine builds on the host entry: packed and scattered end-to-end rates (bench.py's end_to_end leg), twice each, interleaved.
names=("$@")
for rep in 1 2; do for L in "${names[@]}"; do
  lib=variants/libzke_$L.so; [ $L = base ] && lib=zkemail.rs_amd/libzkemail_amd.so
  ZKE_LIB=$lib python bench.py --steps 200 --warmup 20 --no-cpu --no-saturated ${ZKE_AB_ARGS} 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); e=j['end_to_end']; print('$L', 'packed', round(e['value']/1e6,2), 'M/s', e['ms_per_step'], 'scattered', round(e['scattered']['value']/1e6,2), 'M/s', e['scattered']['ms_per_step'])"
done; done
