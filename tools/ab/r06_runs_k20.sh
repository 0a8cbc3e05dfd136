#!/bin/bash
# r06 (final build: VALU-bound): run patterns and the helping threshold at the driver's K = 20 again.   gpurun -- bash tools/ab/r06_runs_k20.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_runs_k20.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0 --steps 20"
for rep in 1 2; do
for runs in "0,1,0,0" "4,2,2,0" "4,2,1,2" "4,1,0,4" "5,1,0,0" "3,2,1,0" "2,1,0,0" "4,3,1,1" "6,2,1,0" "10,1,0,0" "7,6,1,0"; do
  CC4_PERSIST_RUNS=$runs $B 2>/dev/null | line "runs=$runs" >> $OUT
done
for thr in 4 8 32; do CC4_PERSIST_THR=$thr $B 2>/dev/null | line "default runs thr=$thr" >> $OUT; done
done
cat $OUT
