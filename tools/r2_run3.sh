#!/bin/bash
out=gpurun_out/r2i; mkdir -p $out
./tools/ubench/launch_rate 2000 > $out/launch_rate.txt 2>&1; cat $out/launch_rate.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not config4_full_shard and not ragged_and_invalid" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
B="--no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['host_submit_ms'], j['kernels_us_in_flight'], j.get('kernels_us_alone'))" $1; }
python bench.py $B --steps 2000 --warmup 100 > $out/q2000.json 2>$out/q2000.err; val $out/q2000.json
python bench.py $B --steps 20 --warmup 5 > $out/q20.json 2>$out/q20.err; val $out/q20.json
python bench.py $B --steps 20 --warmup 5 > $out/q20b.json 2>$out/q20b.err; val $out/q20b.json
