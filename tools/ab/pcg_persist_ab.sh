#!/bin/bash
# the persistent kernel around the numpy-stream step (k_run_pcg) against four streams of k_step launches: parity probes, then K = 500 / 20
timeout 600 python tools/persist_probe.py 8192 0 2>&1 | tail -7
timeout 600 python tools/persist_probe.py 6000 0 2>&1 | tail -3
for k in 500 20; do for mode in persist streams; do
  if [ $mode = streams ]; then export CC4_PERSIST=0; else unset CC4_PERSIST; fi
  timeout 600 python bench.py --no-alt --no-cpu-baseline --rng pcg64 --steps $k --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$mode K=$k', round(d['value']/1e6,1), 'M', r['kernel'], 'step us', round(r['step_ms']*1e3,1), 'err', d['config']['engine_error_flags'])
"
done; done
