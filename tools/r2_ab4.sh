#!/bin/bash
out=gpurun_out/r2e; mkdir -p $out
B="--steps 2000 --warmup 100 --no-cpu --no-saturated --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'])" $1; }
for sq in "16 20" "20 24" "22 26" "24 28" "28 32" "32 36"; do set -- $sq; GPU_MAX_HW_QUEUES=$2 python bench.py $B --streams $1 > $out/s$1_q$2.json 2>$out/s$1.err; val $out/s$1_q$2.json; done
python bench.py --steps 20 --warmup 5 --no-cpu --no-saturated > $out/d20.json 2>$out/d20.err; val $out/d20.json
python bench.py --steps 20 --warmup 5 --no-cpu --no-saturated > $out/d20b.json 2>$out/d20b.err; val $out/d20b.json
python bench.py --steps 200 --warmup 20 --no-cpu --no-saturated > $out/d200.json 2>$out/d200.err; val $out/d200.json
