#!/bin/bash
# persistent kernel with / without intra-XCD stealing at the tail: parity, then K sweep
export CC4_RUN1=0 CC4_PERSIST_MIN_K=2
export CC4_LIB=$PWD/build_var/steal.so
for n in 8192 6500 16384; do timeout 300 python tools/persist_probe.py $n 2>&1 | tail -5; done
for k in 10 20 32 100 500; do for lib in steal nosteal streams; do
  if [ $lib = streams ]; then export CC4_PERSIST=0; export CC4_LIB=$PWD/build_var/steal.so; else unset CC4_PERSIST; export CC4_LIB=$PWD/build_var/$lib.so; fi
  timeout 300 python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs 8192 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$lib K=$k', round(d['value']/1e6,1), 'M', r['kernel'], 'wall us/region', round(d['ms_per_step']*1e3*$k,1), 'kernel us/region', round(r['step_ms']*1e3*$k,1), 'err', d['config']['engine_error_flags'])
"
done; done
