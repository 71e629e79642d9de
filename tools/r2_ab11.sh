#!/bin/bash
out=gpurun_out/ab11; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_verify.py -m gpu -x -q -k "rounds or quad or corpus or limits" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -3 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
for v in base; do
  for k in 1 2; do python bench.py --no-cpu --no-saturated --steps 2000 --warmup 100 2>$out/$v.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$v', j['value'], j['ms_per_step'], j['kernels_us_alone'])"; done
done
python bench.py --no-cpu --no-saturated --workload c2ed --steps 300 --warmup 40 2>$out/ed.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('c2ed', j['value'], j['ms_per_step'])"
