#!/bin/bash
# parity of the front end after a change + its per-stage instruction counts + the steady-state figure
out=gpurun_out/split; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_verify.py tests/test_gpu_regex.py tests/test_rfc8463_vector.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
bash tools/parse_stage_pmc.sh > $out/stages.txt 2>&1; cat $out/stages.txt
python bench.py --no-cpu --no-saturated --steps 2000 --warmup 100 > $out/bench2000.json 2>$out/bench2000.err || { tail -5 $out/bench2000.err; exit 1; }
python -c "import json; j=json.load(open('$out/bench2000.json')); print('steady', j['value'], j['ms_per_step'], j['kernels_us_alone'])"
