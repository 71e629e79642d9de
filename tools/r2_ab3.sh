#!/bin/bash
out=gpurun_out/r2d; mkdir -p $out
B="--steps 2000 --warmup 100 --no-cpu --no-saturated --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'])" $1; }
ZKE_X_DUMMY_STREAMS=1 python bench.py $B > $out/dummy1.json 2>$out/dummy1.err; val $out/dummy1.json
ZKE_X_DUMMY_STREAMS=2 python bench.py $B > $out/dummy2.json 2>$out/dummy2.err; val $out/dummy2.json
for q in 8 12 16 24 32 40; do GPU_MAX_HW_QUEUES=$q python bench.py $B > $out/q$q.json 2>$out/q$q.err; val $out/q$q.json; done
for q in 16 24 40; do GPU_MAX_HW_QUEUES=$q ZKE_BENCH_MULTI_ENGINE=1 python bench.py $B > $out/mq$q.json 2>$out/mq$q.err; val $out/mq$q.json; done
