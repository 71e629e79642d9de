#!/bin/bash
out=gpurun_out/r2v; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 4"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['kernels_us_in_flight']['rsa_us'], j['kernels_us_alone']['rsa_us'])" $1; }
for v in base vw2 vw4; do
  L=""; [ $v != base ] && L="$PWD/variants/libzke_$v.so"
  ZKE_LIB=$L python bench.py $B --steps 2000 --warmup 100 > $out/${v}_2000.json 2>$out/e.err; val $out/${v}_2000.json
  ZKE_LIB=$L python bench.py $B --steps 20 --warmup 5 > $out/${v}_20.json 2>$out/e.err; val $out/${v}_20.json
  ZKE_LIB=$L python bench.py $B --steps 300 --warmup 40 --workload c2ed > $out/${v}_ed.json 2>$out/e.err; val $out/${v}_ed.json
done
