#!/bin/bash
# persistent kernel (CC4_PERSIST=1) vs four streams for a build: parity probe, then K = 500 / 20 at several sizes
export CC4_LIB=$PWD/$1
export CC4_RUN1=0
python tools/persist_probe.py 8192 2>&1 | tail -6
python tools/persist_probe.py 6000 2>&1 | tail -2
for n in 8192 6144 16384 32768; do for k in 500 20; do for mode in persist streams; do
  if [ $mode = persist ]; then unset CC4_PERSIST; else export CC4_PERSIST=0; fi
  python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), d['roofline'].get('run_kernel'), 'err', d['config']['engine_error_flags'])
"
done; done; done
