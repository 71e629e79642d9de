#!/bin/bash
out=gpurun_out/r2k; mkdir -p $out
B="--no-cpu --no-saturated --steps 3000 --warmup 100 --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], round(1024e6/j['value'],2), 'us/batch host', round(j['host_submit_ms']*1000/j['steps'],2), 'us/step')" $1; }
export ZKE_BENCH_NOCHECK=1
for k in 0 2 3; do ZKE_DEBUG_PARSE_STOP=1 ZKE_DEBUG_SKIP_LAUNCH=$k python bench.py $B > $out/stop1_skip$k.json 2>$out/e.err; val $out/stop1_skip$k.json; done
for k in 0 2 3; do ZKE_DEBUG_PARSE_STOP=7 ZKE_DEBUG_SKIP_LAUNCH=$k python bench.py $B > $out/stop7_skip$k.json 2>$out/e.err; val $out/stop7_skip$k.json; done
ZKE_DEBUG_SKIP_LAUNCH=2 python bench.py $B > $out/full_skip2.json 2>$out/e.err; val $out/full_skip2.json
python bench.py $B > $out/full.json 2>$out/e.err; val $out/full.json
