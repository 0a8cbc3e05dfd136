#!/bin/bash
# A/B of the persistent run kernel (k_run_philox1: K steps of the batch in one launch) against the launch-per-step schedule on four
# streams, same library, same box:  bash tools/persist_ab.sh [lib.so]   (through gpurun)
LIB=${1:-}
[ -n "$LIB" ] && export CC4_LIB=$PWD/$LIB
export CC4_PERSIST_DEBUG=1
python tools/persist_probe.py 2>&1 | tail -12
for n in 8192 16384; do for k in 500 20; do
  for mode in persist streams; do
    if [ $mode = streams ]; then export CC4_NO_PERSIST=1; else unset CC4_NO_PERSIST; fi
    python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), 'launch_ms', round(d['roofline']['launch_ms'],5), 'err', d['config']['engine_error_flags'])
"
  done
done; done
