#!/bin/bash
# Round measurement on the GPU box:  bash tools/measure_round.sh PART [PART ...]   -> gpurun_out/r03/
#   lines      the bench lines (default run, the driver's command, ragged / invalid workloads)
#   c2prof     rocprofv3 kernel summaries of the bench workload in both execution modes + the PMC passes (instructions, HBM traffic)
#   satprof    the chip-filling SHA-256 launches: kernel summary + PMC
#   wlprof     c4shard, c3, c5re, c2ragged: kernel summary + instruction PMC each (dfa / qp / canon kernels get a duration and a count)
#   workloads  every workload's line once
#   stalls     SQ_WAVE_CYCLES split into parked / issue-stalled / issuing per kernel (c2, c5re)
# Every file a figure in README / DESIGN / the bench line cites is copied from there into profiles/ by hand (named r03_*).
set -o pipefail
out=gpurun_out/r03; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --no-e2e"
PMC_I="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES"
line() { python -c "
import json,sys
j=json.loads([l for l in open('$1') if l.startswith('{')][0])
r=j['roofline']
print('$2', j['value'], 'e-mails/s', j['ms_per_step'], 'ms/step; hashed', j.get('hashed_body_GBps'), 'GB/s; in flight', j['kernels_us_in_flight'], 'alone', j['kernels_us_alone'])
for k in ('end_to_end','single_email_latency_us','cpu_baseline'):
    if k in j: print('   ', k, j[k])
if r.get('issue_bound'): print('    issue bound', r['issue_bound'])
"; }
prof() {   # prof NAME bench-args...: kernel summary
  local name=$1; shift
  rm -rf $out/p_$name
  rocprofv3 --kernel-trace --stats -d $out/p_$name -o run --output-format csv -- python bench.py $B "$@" > $out/p_$name.log 2>&1 || { tail -20 $out/p_$name.log; return 1; }
  cp $out/p_$name/run_kernel_stats.csv $out/${name}_kernel_stats.csv
  head -8 $out/${name}_kernel_stats.csv | cut -c1-220
}
pmci_named() {   # like pmci, the counter list taken from the caller's PMC_I, the summary named NAME_pmc.json
  local name=$1; shift
  rm -rf $out/pi_$name
  rocprofv3 --pmc $PMC_I -d $out/pi_$name -o runc --output-format csv -- python bench.py $B "$@" > $out/pi_$name.log 2>&1 || { tail -5 $out/pi_$name.log; return 1; }
  python tools/pmc_summary.py $out/pi_$name > $out/${name}_pmc.json; cat $out/${name}_pmc.json
}
pmci() {   # pmci NAME bench-args...: instruction counters, per kernel and launch
  local name=$1; shift
  rm -rf $out/pi_$name
  rocprofv3 --pmc $PMC_I -d $out/pi_$name -o runc --output-format csv -- python bench.py $B "$@" > $out/pi_$name.log 2>&1 || { tail -5 $out/pi_$name.log; return 1; }
  python tools/pmc_summary.py $out/pi_$name > $out/${name}_instr_pmc.json; cat $out/${name}_instr_pmc.json
}
for part in "$@"; do case $part in
stalls)   # where a wave's cycles go, per kernel: parked (s_waitcnt / barrier), issue-stalled, issuing — the guide's disjoint split of SQ_WAVE_CYCLES
  for W in c2 c5re; do
    PMC_I="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" pmci_named stalls_$W --workload $W --steps 6 --warmup 3 --streams 1 --alone-steps 0 || exit 1
  done
  ;;
lines)
  timeout -k 10 500 python bench.py > $out/bench_line.json 2> $out/bench_line.err || { tail -20 $out/bench_line.err; exit 1; }
  line $out/bench_line.json default
  timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_line.json 2> $out/bench_driver_line.err || { tail -20 $out/bench_driver_line.err; exit 1; }
  line $out/bench_driver_line.json driver
  for W in c2ragged c2inv; do
    timeout -k 10 500 python bench.py --workload $W --steps 1000 --warmup 40 --no-saturated > $out/line_$W.json 2> $out/line_$W.err || { tail -20 $out/line_$W.err; exit 1; }
    line $out/line_$W.json $W
  done
  ;;
c2prof)
  prof bench_c2_inflight --steps 2000 --warmup 100 --alone-steps 0 || exit 1
  prof bench_c2_streams1 --steps 250 --warmup 20 --streams 1 --alone-steps 0 || exit 1
  pmci c2 --steps 8 --warmup 4 --streams 1 --alone-steps 0 || exit 1
  rm -rf $out/p_fetch $out/p_write
  rocprofv3 --pmc FETCH_SIZE -d $out/p_fetch -o runc --output-format csv -- python bench.py $B --steps 10 --warmup 4 --streams 1 --alone-steps 0 > $out/p_fetch.log 2>&1 || { tail -5 $out/p_fetch.log; exit 1; }
  rocprofv3 --pmc WRITE_SIZE -d $out/p_write -o runc --output-format csv -- python bench.py $B --steps 10 --warmup 4 --streams 1 --alone-steps 0 > $out/p_write.log 2>&1 || { tail -5 $out/p_write.log; exit 1; }
  python tools/sha_traffic.py $out/p_fetch $out/p_write > $out/c2_sha_pmc.json && cat $out/c2_sha_pmc.json
  rm -rf $out/p_driver
  rocprofv3 --kernel-trace --stats -d $out/p_driver -o run --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 > $out/p_driver.log 2>&1 || { tail -20 $out/p_driver.log; exit 1; }
  cp $out/p_driver/run_kernel_stats.csv $out/bench_driver_kernel_stats.csv
  ;;
satprof)
  rm -rf $out/p_sat $out/pi_sat $out/p_sat_fetch $out/p_sat_write
  S="--no-cpu --no-e2e --steps 20 --warmup 5 --alone-steps 0"
  rocprofv3 --kernel-trace --stats -d $out/p_sat -o run --output-format csv -- python bench.py $S > $out/p_sat.log 2>&1 || { tail -20 $out/p_sat.log; exit 1; }
  grep -E "Name|sha256" $out/p_sat/run_kernel_stats.csv > $out/sha_saturated_kernel_stats.csv; cat $out/sha_saturated_kernel_stats.csv | cut -c1-220
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d $out/pi_sat -o runc --output-format csv -- python bench.py $S > $out/pi_sat.log 2>&1 || { tail -5 $out/pi_sat.log; exit 1; }
  rocprofv3 --pmc FETCH_SIZE -d $out/p_sat_fetch -o runc --output-format csv -- python bench.py $S > $out/p_sat_fetch.log 2>&1 || { tail -5 $out/p_sat_fetch.log; exit 1; }
  rocprofv3 --pmc WRITE_SIZE -d $out/p_sat_write -o runc --output-format csv -- python bench.py $S > $out/p_sat_write.log 2>&1 || { tail -5 $out/p_sat_write.log; exit 1; }
  python tools/sat_pmc.py $out/pi_sat $out/p_sat_fetch $out/p_sat_write > $out/sha_saturated_pmc.json && cat $out/sha_saturated_pmc.json
  ;;
wlprof)
  for W in c4shard c3 c5re c2ragged; do
    S=400; [ $W = c4shard ] && S=60
    prof bench_$W --workload $W --steps $S --warmup 20 --alone-steps 0 || exit 1
    pmci $W --workload $W --steps 6 --warmup 3 --streams 1 --alone-steps 0 || exit 1
  done
  ;;
workloads)
  : > $out/workloads.txt
  for W in c2 c2inv c2ragged c3 c5 c5re c2ed c4shard; do
    S=1000; [ $W = c4shard ] && S=200; [ $W = c2ed ] && S=300
    timeout -k 10 500 python bench.py --workload $W --steps $S --warmup 40 $B > $out/wl_$W.json 2>$out/wl_$W.err || { tail -5 $out/wl_$W.err; exit 1; }
    line $out/wl_$W.json $W >> $out/workloads.txt
  done
  for b in 4096 8192; do
    timeout -k 10 500 python bench.py --batch $b --steps 500 --warmup 40 $B > $out/wl_b$b.json 2>$out/wl_b$b.err || { tail -5 $out/wl_b$b.err; exit 1; }
    line $out/wl_b$b.json "c2 at $b per batch" >> $out/workloads.txt
  done
  cat $out/workloads.txt
  ;;
*) echo "unknown part $part"; exit 2;;
esac; done
