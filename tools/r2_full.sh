#!/bin/bash
out=gpurun_out/r2m; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -8 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1 || { tail -5 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
python bench.py --steps 20 --warmup 5 > $out/bench20.json 2>$out/bench20.err || { tail -5 $out/bench20.err; exit 1; }
python -c "import json; j=json.load(open('$out/bench20.json')); print(j['value'], j['ms_per_step'], j['roofline']['achieved'], j['cpu_baseline']['value'])"
python bench.py --no-cpu --no-saturated --steps 2000 --warmup 100 > $out/bench2000.json 2>$out/bench2000.err || { tail -5 $out/bench2000.err; exit 1; }
python -c "import json; j=json.load(open('$out/bench2000.json')); print('steady', j['value'], j['ms_per_step'], j['kernels_us_alone'])"
