#!/bin/bash
# A/B: the front end launched as workgroups of W waves (ZKE_PARSE_WG_WAVES; variants/libzke_wg8.so, wg4) — still one e-mail per
# wavefront; W sets how much of a CU the front ends of a burst can occupy (W = 8: 16 waves per CU, 192 registers per SIMD and
# 44 KB of LDS always left for the hash / modexp launches of earlier batches).  -> gpurun_out/r2wg/summary.txt
set -o pipefail
out=gpurun_out/r2wg; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 4"
variants=("$@")
: > $out/summary.txt
for v in "${variants[@]}"; do
  ZKE_LIB=$PWD/variants/libzke_$v.so timeout -k 10 300 python -m pytest tests/test_gpu_verify.py -m gpu -x -q > $out/tests_$v.log 2>&1 || { tail -20 $out/tests_$v.log; exit 1; }
  echo "$v: $(tail -1 $out/tests_$v.log)" | tee -a $out/summary.txt
done
for rep in 1 2; do for v in default "${variants[@]}"; do
  if [ $v = default ]; then unset ZKE_LIB; else export ZKE_LIB=$PWD/variants/libzke_$v.so; fi
  for st in "2000 100" "20 5" "20 5"; do set -- $st
    timeout -k 10 200 python bench.py $B --steps $1 --warmup $2 2>$out/$v.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v steps $1:', round(j['value']/1e6,2), 'M e-mails/s', j['ms_per_step'], 'ms/step, front end alone', j['kernels_us_alone']['parse_us'], 'in flight', j['kernels_us_in_flight']['parse_us'], j['kernels_us_in_flight']['sha_us'])" | tee -a $out/summary.txt || { tail -3 $out/$v.err; }
  done; done; done
