#!/bin/bash
# The N > 1 flow on one GPU (forced through RCCL): the burst three times with the tail's time stamps, then a kernel trace of it.
set -o pipefail
out=gpurun_out/distprof; mkdir -p $out
export TMPDIR=/tmp
B="--no-cpu --no-saturated --alone-steps 0"
export ZKE_BENCH_FORCE_DIST=1 ZKE_BENCH_TAIL_TIMES=1
: > $out/summary.txt
for tag in a b c; do
  timeout -k 10 300 python bench.py $B --gpus 1 --steps 20 --warmup 5 > $out/dist_$tag.json 2> $out/dist_$tag.err || { tail -20 $out/dist_$tag.err; exit 1; }
  echo "dist steps 20: $(grep -o '"value": [0-9.]*' $out/dist_$tag.json) $(grep tail_times $out/dist_$tag.err)" | tee -a $out/summary.txt
done
timeout -k 10 300 python bench.py $B --steps 2000 --warmup 100 > $out/dist_s.json 2> $out/dist_s.err || { tail -20 $out/dist_s.err; exit 1; }
echo "dist steps 2000: $(grep -o '"value": [0-9.]*' $out/dist_s.json) $(grep tail_times $out/dist_s.err)" | tee -a $out/summary.txt
rm -rf $out/p
rocprofv3 --kernel-trace --stats -d $out/p -o run --output-format csv -- python bench.py $B --steps 200 --warmup 20 > $out/p.log 2>&1 || { tail -20 $out/p.log; exit 1; }
find $out/p -name '*kernel_stats.csv' | head -1 | xargs -I{} sh -c 'head -12 {} | cut -c1-220'
