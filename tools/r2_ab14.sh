#!/bin/bash
# front end compiled for 6 (default) / 5 / 4 waves per SIMD
out=gpurun_out/ab14; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[2], j['value'], j['ms_per_step'], j['host_submit_ms'])" $1 "$2"; }
run() { # name env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py $B --steps 2000 --warmup 100 > $out/${name}_s.json 2>$out/e.err || { tail -5 $out/e.err; exit 1; }; val $out/${name}_s.json "$name steady"
  env "$@" timeout -k 10 300 python bench.py $B --steps 20 --warmup 5 > $out/${name}_b.json 2>$out/e.err || { tail -5 $out/e.err; exit 1; }; val $out/${name}_b.json "$name burst"
}
for rep in 1 2; do
run pw6 X=1
run pw5 ZKE_LIB=variants/libzke_pw5.so
run pw4 ZKE_LIB=variants/libzke_pw4.so
done
for W in c3 c5 c4shard; do
  S=1000; [ $W = c4shard ] && S=200
  for v in pw6 pw5; do
    L=zkemail.rs_amd/libzkemail_amd.so; [ $v = pw5 ] && L=variants/libzke_pw5.so
    ZKE_LIB=$L timeout -k 10 300 python bench.py $B --workload $W --steps $S --warmup 40 > $out/${v}_$W.json 2>$out/e.err || { tail -5 $out/e.err; exit 1; }; val $out/${v}_$W.json "$v $W"
  done
done
