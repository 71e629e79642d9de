#!/bin/bash
# Session check on the GPU box: the whole GPU suite, then the driver's bench command and the default one.  -> gpurun_out/s4/
set -o pipefail
out=gpurun_out/s4; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/tests_gpu.log 2>&1 || { tail -30 $out/tests_gpu.log; exit 1; }
tail -2 $out/tests_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1 || { tail -20 $out/smoke.log; exit 1; }
tail -1 $out/smoke.log
for i in 1 2 3; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_$i.json 2> $out/bench_driver_$i.err || { tail -20 $out/bench_driver_$i.err; exit 1; }
done
timeout -k 10 400 python bench.py > $out/bench_line.json 2> $out/bench_line.err || { tail -20 $out/bench_line.err; exit 1; }
python - <<'PY'
import json
for f in ('bench_driver_1','bench_driver_2','bench_driver_3','bench_line'):
    j=json.load(open('gpurun_out/s4/'+f+'.json')); print(f, j['value'], j['ms_per_step'], j['roofline']['frac'], j['cpu_baseline']['value'])
PY
