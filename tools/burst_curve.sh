#!/bin/bash
# T(K): the timed region of bench.py for K = 1 .. 160 steps (device-resident, 22 slots), three runs each — the fixed part
# (ramp + the last batch's chain under load) and the marginal cost per batch of the burst the driver measures (K = 20).
for K in 1 2 5 10 20 40 80 160; do for rep in 1 2 3; do
  python bench.py --steps $K --warmup 5 --no-cpu --no-saturated --no-e2e 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('K', $K, 'total_us', round(j['ms_per_step']*1e3*$K,1), 'per_step_us', round(j['ms_per_step']*1e3,1), 'M/s', round(j['value']/1e6,2))"
done; done
