#!/bin/bash
# A/B of the round-2 engine (one engine, S slots) against the round-1 tree (old_r01/: 20 engines), same box, same run.
out=gpurun_out/r2b; mkdir -p $out
B="--steps 2000 --warmup 100 --no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'])" $1; }
python old_r01/bench.py $B > $out/old.json 2>$out/old.err; val $out/old.json
for s in 12 16 20 24; do python bench.py $B --alone-steps 0 --streams $s > $out/new_s$s.json 2>$out/new_s$s.err; val $out/new_s$s.json; done
for r in 4 20; do ZKE_KEY_CACHE_REPLICAS=$r python bench.py $B --alone-steps 0 > $out/new_rep$r.json 2>$out/new_rep$r.err; val $out/new_rep$r.json; done
python old_r01/bench.py $B > $out/old2.json 2>$out/old2.err; val $out/old2.json
