#!/bin/bash
# The driver's burst (--steps 20 --warmup 5) against the number of submission slots.
out=gpurun_out/r2p; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['host_submit_ms'])" $1; }
for S in 4 5 6 7 8 10 12 14 16 18 20 22; do
  for k in 1 2; do timeout -k 10 200 python bench.py $B --steps 20 --warmup 5 --streams $S > $out/s${S}_$k.json 2>$out/e.err || { tail -5 $out/e.err; exit 1; }; val $out/s${S}_$k.json; done
done
