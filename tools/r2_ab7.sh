#!/bin/bash
out=gpurun_out/r2u; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'])" $1; }
python bench.py $B --steps 2000 --warmup 100 > $out/base.json 2>$out/e.err; val $out/base.json
ZKE_SHA_TILE=64 python bench.py $B --steps 2000 --warmup 100 > $out/tile64.json 2>$out/e.err; val $out/tile64.json
ZKE_SHA_TILE=256 python bench.py $B --steps 2000 --warmup 100 > $out/tile256.json 2>$out/e.err; val $out/tile256.json
python bench.py $B --steps 2000 --warmup 100 --streams 22 > $out/s22.json 2>$out/e.err; val $out/s22.json
python bench.py $B --steps 2000 --warmup 100 --streams 21 > $out/s21.json 2>$out/e.err; val $out/s21.json
ZKE_SHA_TILE=64 python bench.py $B --steps 20 --warmup 5 > $out/tile64_20.json 2>$out/e.err; val $out/tile64_20.json
python bench.py $B --steps 20 --warmup 5 > $out/base_20.json 2>$out/e.err; val $out/base_20.json
python bench.py $B --steps 2000 --warmup 100 > $out/base2.json 2>$out/e.err; val $out/base2.json
