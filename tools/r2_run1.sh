#!/bin/bash
out=gpurun_out/r2f; mkdir -p $out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -15 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
B="--no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['kernels_us_in_flight'], j.get('kernels_us_alone'))" $1; }
python bench.py $B --steps 2000 --warmup 100 > $out/wave2000.json 2>$out/wave2000.err; val $out/wave2000.json
ZKE_RSA_QUAD_MIN=0 python bench.py $B --steps 2000 --warmup 100 > $out/quad2000.json 2>$out/quad2000.err; val $out/quad2000.json
python bench.py $B --steps 20 --warmup 5 > $out/wave20.json 2>$out/wave20.err; val $out/wave20.json
ZKE_RSA_QUAD_MIN=0 python bench.py $B --steps 20 --warmup 5 > $out/quad20.json 2>$out/quad20.err; val $out/quad20.json
