#!/bin/bash
# kernel-trace of the driver's burst, as a per-batch timeline
out=gpurun_out/tl; mkdir -p $out; export TMPDIR=/tmp
rm -rf $out/p
rocprofv3 --kernel-trace -d $out/p -o run --output-format csv -- python bench.py --no-cpu --no-saturated --alone-steps 0 --steps 20 --warmup 5 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
python tools/burst_timeline.py $out/p/run_kernel_trace.csv 22 5 20 > $out/timeline.txt; cat $out/timeline.txt
