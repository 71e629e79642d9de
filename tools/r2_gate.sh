#!/bin/bash
out=gpurun_out/r2s; mkdir -p $out
B="--no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['host_submit_ms'])" $1; }
for g in 0 4 6 8 10 12; do
ZKE_FRONT_GATE=$g python bench.py $B --steps 20 --warmup 5 > $out/g${g}_20.json 2>$out/e.err; val $out/g${g}_20.json
done
for g in 0 6 10; do
ZKE_FRONT_GATE=$g python bench.py $B --steps 2000 --warmup 100 > $out/g${g}_2000.json 2>$out/e.err; val $out/g${g}_2000.json
done
for g in 6 10; do ZKE_FRONT_GATE=$g python bench.py $B --steps 40 --warmup 5 > $out/g${g}_40.json 2>$out/e.err; val $out/g${g}_40.json; done
