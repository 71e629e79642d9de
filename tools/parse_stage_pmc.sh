#!/bin/bash
# VALU / SALU / LDS instructions of the front-end kernel truncated after stage k (ZKE_DEBUG_PARSE_STOP=k): where the
# parser's instructions come from.  bash tools/parse_stage_pmc.sh  (GPU box; writes gpurun_out/pstage_k/)
export TMPDIR=/tmp
for k in 1 2 3 4 5 6 7 0; do
  rm -rf gpurun_out/pstage_$k
  ZKE_DEBUG_PARSE_STOP=$k ZKE_BENCH_NOCHECK=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR -d gpurun_out/pstage_$k -o runc --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu --no-saturated --streams 1 > gpurun_out/pstage_$k.log 2>&1
done
