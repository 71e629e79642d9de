#!/bin/bash
# ONE test of the hipGraph replay path (opt-in, ZKE_GRAPHS=1) — see DESIGN.md §6.  Stops at the first failure.
out=gpurun_out/r2l; mkdir -p $out
set -o pipefail
export ZKE_GRAPHS=1
timeout -k 10 300 python -m pytest tests/test_gpu_verify.py -m gpu -x -q -k "device_resident_entry" > $out/pytest_graphs.log 2>&1; rc=$?
tail -5 $out/pytest_graphs.log
if [ $rc -ne 0 ] || grep -q "Memory access fault" $out/pytest_graphs.log; then echo "GRAPH TEST FAILED rc=$rc"; exit 1; fi
B="--no-cpu --no-saturated --alone-steps 0"
val() { python -c "import json,sys; j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); print(sys.argv[1], j['value'], round(1024e6/j['value'],2), 'us/batch host', round(j['host_submit_ms']*1000/j['steps'],2), 'us/step')" $1; }
timeout -k 10 200 python bench.py $B --steps 2000 --warmup 100 > $out/g2000.json 2>$out/g2000.err || { tail -5 $out/g2000.err; exit 1; }
grep -q "Memory access fault" $out/g2000.err && { echo FAULT; exit 1; }
val $out/g2000.json
timeout -k 10 200 python bench.py $B --steps 20 --warmup 45 > $out/g20.json 2>$out/g20.err || { tail -5 $out/g20.err; exit 1; }
val $out/g20.json
ZKE_BENCH_NOCHECK=1 ZKE_DEBUG_PARSE_STOP=1 timeout -k 10 200 python bench.py $B --steps 3000 --warmup 100 > $out/g_stop1.json 2>$out/g_stop1.err || { tail -5 $out/g_stop1.err; exit 1; }
val $out/g_stop1.json
