#!/bin/bash
# experiment: HIP runtime knobs against the per-launch cost (steady state, 2 000 steps, and the empty pipeline)
out=gpurun_out/r2env; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 0"
run() {  # label, env assignments...
  local label="$1"; shift
  ( for kv in "$@"; do export "$kv"; done
    timeout -k 10 200 python bench.py $B --steps 2000 --warmup 100 2>$out/$label.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$label full :', round(j['value']/1e6,2), 'M e-mails/s', j['ms_per_step'], 'ms/step')"
    export ZKE_DEBUG_PARSE_STOP=1 ZKE_BENCH_NOCHECK=1
    timeout -k 10 200 python bench.py $B --steps 3000 --warmup 100 2>>$out/$label.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$label empty:', j['ms_per_step'], 'ms/step')" ) || echo "$label failed"
}
run default
run dev_kernarg0 HIP_FORCE_DEV_KERNARG=0
run dev_kernarg1 HIP_FORCE_DEV_KERNARG=1
run opt_flush0 AMD_OPT_FLUSH=0
run sysscope0 ROC_SYSTEM_SCOPE_SIGNAL=0
run kernarg_copy_opt0 DEBUG_HIP_KERNARG_COPY_OPT=0
run hdp_wa0 DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run fgs_kernarg0 ROC_USE_FGS_KERNARG=0
run dyn_queues1 DEBUG_HIP_DYNAMIC_QUEUES=1
run aql_size ROC_AQL_QUEUE_SIZE=4096
