#!/bin/bash
# Sweep: batch size x launches per step (episode groups on separate streams, CC4_GROUPS) x counter-mode kernel (CC4_PHILOX_LEAN).
# usage: bash tools/groups_sweep.sh [rng]      (run through gpurun)
RNG=${1:-philox}
for n in 1024 2048 3072 4096 6144 8192 16384; do for g in 1 2 3 4; do for lean in 0 1; do
  if [ $RNG = pcg64 ] && [ $lean = 1 ]; then continue; fi
  CC4_GROUPS=$g CC4_PHILOX_LEAN=$lean python bench.py --rng $RNG --no-alt --no-cpu-baseline --total-envs $n --min-seconds 0.25 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$RNG n=$n groups=$g lean=$lean', d['roofline']['kernel'], round(d['value']/1e6,1), 'M  step_us', round(d['ms_per_step']*1e3,2), 'err', d['config']['engine_error_flags'])
"
done; done; done
