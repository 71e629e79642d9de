#!/bin/bash
out=gpurun_out/r2n; mkdir -p $out
B="--no-cpu --no-saturated --steps 2000 --warmup 100 --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'])" $1; }
python bench.py $B > $out/base.json 2>$out/e.err; val $out/base.json
for v in p00 p30 p11 pp2; do ZKE_LIB=$PWD/variants/libzke_$v.so python bench.py $B > $out/$v.json 2>$out/e.err; val $out/$v.json; done
for sq in "18 22" "22 26" "24 28"; do set -- $sq; GPU_MAX_HW_QUEUES=$2 python bench.py $B --streams $1 > $out/s$1.json 2>$out/e.err; val $out/s$1.json; done
python bench.py $B > $out/base2.json 2>$out/e.err; val $out/base2.json
for v in p00 pp2; do ZKE_LIB=$PWD/variants/libzke_$v.so python bench.py --no-cpu --no-saturated --steps 20 --warmup 5 > $out/${v}_20.json 2>$out/e.err; val $out/${v}_20.json; done
