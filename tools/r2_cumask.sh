#!/bin/bash
# experiment: every slot's launches confined to 32 of the 256 CUs (hipExtStreamCreateWithCUMask), two bit layouts
out=gpurun_out/r2cu; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 0"
for m in 0 1 2; do
  if [ $m = 0 ]; then unset ZKE_X_CU_MASK; else export ZKE_X_CU_MASK=$m; fi
  for st in "2000 100" "20 5"; do set -- $st
    timeout -k 10 200 python bench.py $B --steps $1 --warmup $2 2>$out/m${m}_$1.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('mask mode $m steps $1:', round(j['value']/1e6,2), 'M e-mails/s', j['ms_per_step'], 'ms/step', j.get('kernels_us_in_flight'))" || { tail -5 $out/m${m}_$1.err; exit 1; }
  done
done
export ZKE_DEBUG_PARSE_STOP=1 ZKE_BENCH_NOCHECK=1
for m in 0 1 2; do
  if [ $m = 0 ]; then unset ZKE_X_CU_MASK; else export ZKE_X_CU_MASK=$m; fi
  timeout -k 10 200 python bench.py $B --steps 3000 --warmup 100 2>$out/e${m}.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('empty pipeline, mask mode $m:', j['ms_per_step'], 'ms/step')" || { tail -5 $out/e${m}.err; exit 1; }
done
