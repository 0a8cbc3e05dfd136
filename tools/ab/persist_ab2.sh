#!/bin/bash
# A/B of a build's persistent run kernel (CC4_PERSIST=1) against its launch-per-step schedule: bash tools/persist_ab2.sh build_var/x.so
export CC4_LIB=$PWD/$1
CC4_PERSIST_DEBUG=1 python tools/persist_probe.py 2>&1 | tail -7
for n in 8192 4096 16384; do for k in 500 20; do for mode in persist streams; do
  if [ $mode = persist ]; then unset CC4_PERSIST; else export CC4_PERSIST=0; fi
  python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), d['roofline'].get('run_kernel'), 'err', d['config']['engine_error_flags'])
"
done; done; done
