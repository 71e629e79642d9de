#!/bin/bash
out=gpurun_out/r2c; mkdir -p $out
B="--steps 2000 --warmup 100 --no-cpu --no-saturated --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'])" $1; }
ZKE_BENCH_TORCH_STREAMS=1 python bench.py $B > $out/tstreams.json 2>$out/tstreams.err; val $out/tstreams.json
ZKE_BENCH_MULTI_ENGINE=1 python bench.py $B > $out/multi.json 2>$out/multi.err; val $out/multi.json
ZKE_BENCH_MULTI_ENGINE=1 ZKE_BENCH_TORCH_STREAMS=1 python bench.py $B > $out/multi_t.json 2>$out/multi_t.err; val $out/multi_t.json
ZKE_STREAM_PRIO=0 python bench.py $B > $out/prio0.json 2>$out/prio0.err; val $out/prio0.json
python bench.py $B > $out/base.json 2>$out/base.err; val $out/base.json
