#!/bin/bash
out=gpurun_out/r2o; mkdir -p $out
B="--no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['host_submit_ms'])" $1; }
for k in 1 2 3; do python bench.py $B --steps 20 --warmup 5 > $out/b20_$k.json 2>$out/e.err; val $out/b20_$k.json; done
python bench.py $B --steps 20 --warmup 0 > $out/b20_w0.json 2>$out/e.err; val $out/b20_w0.json
python bench.py $B --steps 20 --warmup 40 > $out/b20_w40.json 2>$out/e.err; val $out/b20_w40.json
python bench.py $B --steps 40 --warmup 5 > $out/b40.json 2>$out/e.err; val $out/b40.json
python bench.py $B --steps 100 --warmup 5 > $out/b100.json 2>$out/e.err; val $out/b100.json
python bench.py $B --steps 2000 --warmup 100 > $out/b2000.json 2>$out/e.err; val $out/b2000.json
