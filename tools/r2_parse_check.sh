#!/bin/bash
out=gpurun_out/r2q; mkdir -p $out
timeout -k 10 400 python -m pytest tests/test_gpu_verify.py tests/test_gpu_regex.py tests/test_rfc8463_vector.py -m gpu -x -q > $out/tests.log 2>&1 || { tail -30 $out/tests.log; exit 1; }
tail -2 $out/tests.log
bash tools/parse_stage_pmc.sh 2>&1 | tee $out/pstage.txt
B="--no-cpu --no-saturated"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], j['ms_per_step'], j['kernels_us_alone'])" $1; }
python bench.py $B --steps 2000 --warmup 100 > $out/b2000.json 2>$out/e.err; val $out/b2000.json
python bench.py $B --steps 20 --warmup 5 > $out/b20.json 2>$out/e.err; val $out/b20.json
