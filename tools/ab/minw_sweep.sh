#!/bin/bash
# four-wave kernel: register budget (CC4_PHILOX_MINW) x launches per step, at the batch sizes it serves
for n in 1024 2048 4096; do for g in 2 3; do for w in 1 7 8; do
  CC4_GROUPS=$g CC4_PHILOX_LEAN=0 CC4_PHILOX_MINW=$w python bench.py --no-alt --no-cpu-baseline --total-envs $n --min-seconds 0.25 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('n=$n groups=$g minw=$w', d['roofline']['kernel'], round(d['value']/1e6,1), 'M  step_us', round(d['ms_per_step']*1e3,2), 'err', d['config']['engine_error_flags'])
"
done; done; done
