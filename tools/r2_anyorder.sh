#!/bin/bash
# experiment: two workspaces per hardware queue (ZKE_X_SHARE=22: slots 22..43 on the streams of slots 0..21) and the front end
# launched without the queue's barrier bit (ZKE_X_ANYORDER, hipExtAnyOrderLaunch): a batch's front end beside the previous batch's verdict
out=gpurun_out/r2any; mkdir -p $out
B="--no-cpu --no-saturated --alone-steps 0"
run() {  # label, streams, env...
  local label="$1" st="$2"; shift; shift
  ( for kv in "$@"; do export "$kv"; done
    for steps in "2000 100" "20 5"; do set -- $steps
      timeout -k 10 200 python bench.py $B --streams $st --steps $1 --warmup $2 2>$out/$label.err | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$label steps $1:', round(j['value']/1e6,2), 'M e-mails/s', j['ms_per_step'], 'ms/step')"
    done ) || echo "$label failed"
}
run base22 22
run share44 44 GPU_MAX_HW_QUEUES=26 ZKE_X_SHARE=22
run share44_anyorder 44 GPU_MAX_HW_QUEUES=26 ZKE_X_SHARE=22 ZKE_X_ANYORDER=1
run share36_anyorder 36 GPU_MAX_HW_QUEUES=22 ZKE_X_SHARE=18 ZKE_X_ANYORDER=1
run share48_anyorder 48 GPU_MAX_HW_QUEUES=28 ZKE_X_SHARE=24 ZKE_X_ANYORDER=1
