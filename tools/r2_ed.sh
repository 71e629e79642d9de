#!/bin/bash
out=gpurun_out/ed; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_blocks.py tests/test_gpu_verify.py tests/test_rfc8463_vector.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log; tail -5 $out/pytest.log
grep -q "pytest rc=0" $out/pytest.log || exit 1
python bench.py --no-cpu --no-saturated --workload c2ed --steps 300 --warmup 40 2>$out/ed.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('c2ed', j['value'], j['ms_per_step'], j['kernels_us_alone'])"
python bench.py --no-cpu --no-saturated --steps 2000 --warmup 100 2>$out/c2.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('c2', j['value'], j['ms_per_step'], j['kernels_us_alone'])"
