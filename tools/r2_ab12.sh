#!/bin/bash
# verdict launch (round loop, next round as a call) at 1 / 2 waves per SIMD
out=gpurun_out/ab12; mkdir -p $out
echo "pytest rc=0" > $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
for v in vw2 vw3 vw4 vw2 vw3 vw4; do
  L="variants/libzke_$v.so"
  ZKE_LIB=$L python bench.py --no-cpu --no-saturated --workload c2ed --steps 300 --warmup 40 2>$out/$v.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v c2ed', j['value'], j['ms_per_step'], j['kernels_us_alone']['rsa_us'])"
  ZKE_LIB=$L python bench.py --no-cpu --no-saturated --steps 2000 --warmup 100 2>$out/$v.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v c2', j['value'], j['ms_per_step'], j['kernels_us_alone']['rsa_us'])"
done
