#!/bin/bash
# more launches per step than three: does the runtime's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) set the cliff at four streams?
for q in 4 8; do for n in 1024 2048 8192; do for g in 3 4 5 6 8; do
  GPU_MAX_HW_QUEUES=$q CC4_GROUPS=$g python bench.py --no-alt --no-cpu-baseline --total-envs $n --min-seconds 0.25 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('hwq=$q n=$n groups=$g', d['roofline']['kernel'], round(d['value']/1e6,1), 'M  step_us', round(d['ms_per_step']*1e3,2))
"
done; done; done
