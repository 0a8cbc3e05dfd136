#!/bin/bash
# batch sizes between the kernels' home ranges: what cc4_run_random_steps picks (auto) against per-step launches (all one-launch forms off)
for n in 3072 5500; do timeout 300 python tools/persist_probe.py $n 2>&1 | tail -2; done
timeout 300 python tools/persist_probe.py 5000 0 2>&1 | tail -2
for n in 2560 3072 3584 5500 6000; do for mode in auto perstep; do
  if [ $mode = perstep ]; then export CC4_PERSIST=0 CC4_RUN1=0 CC4_MULTISTEP=0; else unset CC4_PERSIST CC4_RUN1 CC4_MULTISTEP; fi
  timeout 300 python bench.py --no-alt --no-cpu-baseline --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$mode n=$n', round(d['value']/1e6,1), 'M', r['kernel'], 'step us', round(r['step_ms']*1e3,1), 'err', d['config']['engine_error_flags'])
"
done; done
for mode in auto perstep; do
  if [ $mode = perstep ]; then export CC4_PERSIST=0; else unset CC4_PERSIST; fi
  timeout 300 python bench.py --no-alt --no-cpu-baseline --rng pcg64 --total-envs 5000 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('pcg $mode n=5000', round(d['value']/1e6,1), 'M', r['kernel'], 'step us', round(r['step_ms']*1e3,1), 'err', d['config']['engine_error_flags'])
"
done
