#!/bin/bash
out=gpurun_out/r2r; mkdir -p $out
B="--no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['kernels_us_in_flight']['rsa_us'])" $1; }
for k in 1 2; do
python bench.py $B --steps 2000 --warmup 100 > $out/base_$k.json 2>$out/e.err; val $out/base_$k.json
ZKE_LIB=$PWD/variants/libzke_noed.so python bench.py $B --steps 2000 --warmup 100 > $out/noed_$k.json 2>$out/e.err; val $out/noed_$k.json
python bench.py $B --steps 20 --warmup 5 > $out/base20_$k.json 2>$out/e.err; val $out/base20_$k.json
ZKE_LIB=$PWD/variants/libzke_noed.so python bench.py $B --steps 20 --warmup 5 > $out/noed20_$k.json 2>$out/e.err; val $out/noed20_$k.json
done
