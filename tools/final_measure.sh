#!/bin/bash
# Round-end measurement on the GPU box: full GPU suite, the default bench line, the rocprofv3 kernel summary of the
# bench workload (one batch at a time, as the per-kernel times in the bench line) and the SHA launch's HBM traffic.
# bash tools/final_measure.sh   -> gpurun_out/{final_tests.log,bench_full.log,p_stats/,p_c2_fetch/,p_c2_write/}
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -40 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
timeout -k 10 400 python bench.py > gpurun_out/bench_full.log 2> gpurun_out/bench_full.err || { tail -20 gpurun_out/bench_full.err; exit 1; }
grep -o '"value": [0-9.]*' gpurun_out/bench_full.log | head -1
export TMPDIR=/tmp
rm -rf gpurun_out/p_stats gpurun_out/p_c2_fetch gpurun_out/p_c2_write
rocprofv3 --kernel-trace --stats -d gpurun_out/p_stats -o run --output-format csv -- python bench.py --steps 250 --warmup 20 --no-cpu --no-saturated --streams 1 > gpurun_out/p_stats.log 2>&1 || { tail -20 gpurun_out/p_stats.log; exit 1; }
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/p_c2_fetch -o runc --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu --no-saturated --streams 1 > gpurun_out/p_c2_fetch.log 2>&1 || { tail -20 gpurun_out/p_c2_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/p_c2_write -o runc --output-format csv -- python bench.py --steps 10 --warmup 2 --no-cpu --no-saturated --streams 1 > gpurun_out/p_c2_write.log 2>&1 || { tail -20 gpurun_out/p_c2_write.log; exit 1; }
python tools/sha_traffic.py gpurun_out/p_c2_fetch gpurun_out/p_c2_write > gpurun_out/c2_sha_pmc.json && cat gpurun_out/c2_sha_pmc.json
head -6 gpurun_out/p_stats/run_kernel_stats.csv
