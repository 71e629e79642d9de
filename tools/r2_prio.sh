#!/bin/bash
# slot priorities: slot k at priority level k / G (earlier slots first), burst and steady state
for g in 0 8 11 4; do
  for st in "20 5" "2000 100"; do
    set -- $st
    E=""; [ $g != 0 ] && E="ZKE_X_SLOT_PRIO_GROUPS=$g"
    env $E python bench.py --no-cpu --no-saturated --steps $1 --warmup $2 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('groups $g steps $1', j['value'], j['ms_per_step'])"
  done
done
