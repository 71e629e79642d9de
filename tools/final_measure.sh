#!/bin/bash
# Round-end measurement on the GPU box (bash tools/final_measure.sh): the bench lines, the rocprofv3 kernel summaries of
# the bench workload in both execution modes (S batches in flight = the timed region's mode; one batch at a time), the PMC
# instruction counts and the hash / modexp launch's HBM traffic, every other workload once.  -> gpurun_out/final/
set -o pipefail
out=gpurun_out/final; mkdir -p $out
export TMPDIR=/tmp
python -m zkemail_rs_amd.build > /dev/null
timeout -k 10 300 python -m pytest tests/test_gpu_verify.py tests/test_gpu_bench_contract.py -m gpu -x -q > $out/tests_quick.log 2>&1 || { tail -30 $out/tests_quick.log; exit 1; }
tail -2 $out/tests_quick.log
timeout -k 10 400 python bench.py > $out/bench_line.json 2> $out/bench_line.err || { tail -20 $out/bench_line.err; exit 1; }
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_line.json 2> $out/bench_driver_line.err || { tail -20 $out/bench_driver_line.err; exit 1; }
python -c "
import json
for f in ('bench_line','bench_driver_line'):
    j=json.load(open('$out/'+f+'.json')); print(f, j['value'], j['ms_per_step'], j['roofline']['achieved'], j['roofline']['per_launch']['alone'], j['cpu_baseline']['value'])"
B="--no-cpu --no-saturated"
rm -rf $out/p_inflight $out/p_alone $out/p_instr $out/p_fetch $out/p_write
rocprofv3 --kernel-trace --stats -d $out/p_inflight -o run --output-format csv -- python bench.py $B --steps 2000 --warmup 100 --alone-steps 0 > $out/p_inflight.log 2>&1 || { tail -20 $out/p_inflight.log; exit 1; }
rocprofv3 --kernel-trace --stats -d $out/p_alone -o run --output-format csv -- python bench.py $B --steps 250 --warmup 20 --streams 1 --alone-steps 0 > $out/p_alone.log 2>&1 || { tail -20 $out/p_alone.log; exit 1; }
head -6 $out/p_inflight/run_kernel_stats.csv | cut -c1-200; head -6 $out/p_alone/run_kernel_stats.csv | cut -c1-200
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d $out/p_instr -o runc --output-format csv -- python bench.py $B --steps 8 --warmup 4 --streams 1 --alone-steps 0 > $out/p_instr.log 2>&1 || { tail -5 $out/p_instr.log; exit 1; }
python tools/pmc_summary.py $out/p_instr > $out/c2_instr_pmc.json; cat $out/c2_instr_pmc.json
rocprofv3 --pmc FETCH_SIZE -d $out/p_fetch -o runc --output-format csv -- python bench.py $B --steps 10 --warmup 4 --streams 1 --alone-steps 0 > $out/p_fetch.log 2>&1 || { tail -5 $out/p_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d $out/p_write -o runc --output-format csv -- python bench.py $B --steps 10 --warmup 4 --streams 1 --alone-steps 0 > $out/p_write.log 2>&1 || { tail -5 $out/p_write.log; exit 1; }
python tools/sha_traffic.py $out/p_fetch $out/p_write > $out/c2_sha_pmc.json && cat $out/c2_sha_pmc.json
: > $out/workloads.txt
for W in c2 c3 c5 c5re c2ed c4shard; do
  S=1000; [ $W = c4shard ] && S=200; [ $W = c2ed ] && S=300
  timeout -k 10 400 python bench.py --workload $W --steps $S --warmup 40 $B 2>$out/wl_$W.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$W', j['value'], 'e-mails/s', j['ms_per_step'], 'ms/step  in flight', j['kernels_us_in_flight'], ' alone', j['kernels_us_alone'])" >> $out/workloads.txt || { tail -5 $out/wl_$W.err; exit 1; }
done
for b in 4096 8192; do timeout -k 10 400 python bench.py --batch $b --steps 500 --warmup 40 $B 2>$out/wl_b$b.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('c2 at $b per batch', j['value'], 'e-mails/s', j['ms_per_step'], 'ms/step')" >> $out/workloads.txt; done
cat $out/workloads.txt
