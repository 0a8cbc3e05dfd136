#!/bin/bash
# A/B of the multi-step four-wave kernel (k_run_philox: K steps of a small batch in one launch, every block looping over the steps
# of its episode) against the launch-per-step schedule, same library, same box:  bash tools/multistep_ab.sh [lib.so]  (through gpurun)
LIB=${1:-}
[ -n "$LIB" ] && export CC4_LIB=$PWD/$LIB
export CC4_PERSIST_DEBUG=1
python tools/persist_probe.py 1024 2>&1 | tail -9
python tools/persist_probe.py 300 2>&1 | tail -3
for n in 1024 512 1280 2048 256; do for k in 500 20; do
  for mode in multistep launches; do
    if [ $mode = launches ]; then export CC4_MULTISTEP=0; else export CC4_MULTISTEP=1; fi
    CC4_PERSIST_DEBUG= python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), 'launch_ms', round(d['roofline']['launch_ms'],5), d['roofline']['kernel'], 'err', d['config']['engine_error_flags'])
"
  done
done; done
