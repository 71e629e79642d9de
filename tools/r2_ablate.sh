#!/bin/bash
out=gpurun_out/r2h; mkdir -p $out
B="--no-cpu --no-saturated --steps 2000 --warmup 100 --alone-steps 4"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], round(1024e6/j['value'],2), 'us/batch', j['kernels_us_alone'])" $1; }
export ZKE_BENCH_NOCHECK=1
for s in 1 2 4 5 6 7; do ZKE_DEBUG_PARSE_STOP=$s python bench.py $B > $out/stop$s.json 2>$out/stop$s.err; val $out/stop$s.json; done
ZKE_DEBUG_SKIP_RSA=1 python bench.py $B > $out/skiprsa.json 2>$out/skiprsa.err; val $out/skiprsa.json
ZKE_NO_FUSE_CANON=1 python bench.py $B > $out/nofuse.json 2>$out/nofuse.err; val $out/nofuse.json
python bench.py $B > $out/full.json 2>$out/full.err; val $out/full.json
