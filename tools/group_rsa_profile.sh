#!/bin/bash
# Instruction counters and kernel durations of the lane-group RSA kernels (GPU box):
#   bash tools/group_rsa_profile.sh -> gpurun_out/p_b4096_instr.json, p_c5_instr.json, p_b4096_stats/, p_c5_stats/
export TMPDIR=/tmp
set -o pipefail
for W in b4096 c5; do
  if [ $W = b4096 ]; then A="--batch 4096"; else A="--workload c5"; fi
  rm -rf gpurun_out/p_${W}_instr gpurun_out/p_${W}_stats
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU -d gpurun_out/p_${W}_instr -o runc --output-format csv -- python bench.py $A --steps 6 --warmup 2 --no-cpu --no-saturated --streams 1 > gpurun_out/p_${W}_instr.log 2>&1 || { tail -5 gpurun_out/p_${W}_instr.log; exit 1; }
  python tools/pmc_summary.py gpurun_out/p_${W}_instr > gpurun_out/p_${W}_instr.json
  rocprofv3 --kernel-trace --stats -d gpurun_out/p_${W}_stats -o run --output-format csv -- python bench.py $A --steps 60 --warmup 10 --no-cpu --no-saturated --streams 1 > gpurun_out/p_${W}_stats.log 2>&1 || { tail -5 gpurun_out/p_${W}_stats.log; exit 1; }
  head -8 gpurun_out/p_${W}_stats/run_kernel_stats.csv | cut -c1-160
  cat gpurun_out/p_${W}_instr.json | head -60
done
