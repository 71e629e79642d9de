#!/bin/bash
# VALU / SALU / LDS instructions of the front-end kernel truncated after stage k (ZKE_DEBUG_PARSE_STOP=k): where the
# parser's instructions come from.  bash tools/parse_stage_pmc.sh  (GPU box; writes gpurun_out/pstage/)
export TMPDIR=/tmp
mkdir -p gpurun_out/pstage
for k in 1 2 3 4 5 6 7 0; do
  rm -rf gpurun_out/pstage/k$k
  ZKE_DEBUG_PARSE_STOP=$k ZKE_BENCH_NOCHECK=1 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES -d gpurun_out/pstage/k$k -o runc --output-format csv -- python bench.py --steps 6 --warmup 2 --no-cpu --no-saturated --streams 1 --alone-steps 0 > gpurun_out/pstage/k$k.log 2>&1
  python tools/pmc_summary.py gpurun_out/pstage/k$k | python -c "
import json,sys
j=json.load(sys.stdin)['zke::parse_kernel']
print('stop=$k', {k:round(v/1024) for k,v in j.items() if k.startswith('SQ_')})"
done
