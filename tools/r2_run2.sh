#!/bin/bash
out=gpurun_out/r2g; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
B="--no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['host_submit_ms'], j['kernels_us_in_flight'], j.get('kernels_us_alone'))" $1; }
python bench.py $B --steps 2000 --warmup 100 > $out/q2000.json 2>$out/q2000.err; val $out/q2000.json
python bench.py $B --steps 20 --warmup 5 > $out/q20.json 2>$out/q20.err; val $out/q20.json
python bench.py $B --steps 20 --warmup 5 > $out/q20b.json 2>$out/q20b.err; val $out/q20b.json
python bench.py $B --steps 200 --warmup 20 > $out/q200.json 2>$out/q200.err; val $out/q200.json
ZKE_RSA_QUAD=0 python bench.py $B --steps 2000 --warmup 100 > $out/w2000.json 2>$out/w2000.err; val $out/w2000.json
for w in c4shard c5 c3 c5re c2ed; do python bench.py $B --steps 300 --warmup 40 --workload $w > $out/$w.json 2>$out/$w.err; val $out/$w.json; done
