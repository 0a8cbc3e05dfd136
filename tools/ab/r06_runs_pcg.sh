#!/bin/bash
# r06: run patterns of the numpy-stream persistent kernel at the driver's K = 20 (its steps are 2.7 x longer than the counter mode's: a run of 4 is a 350 us
# item in a 1.7 ms call).   gpurun -- bash tools/ab/r06_runs_pcg.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_runs_pcg.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0 --steps 20 --rng pcg64"
for rep in 1 2; do
for runs in "0,1,0,0" "2,1,0,0" "1,1,0,0" "3,1,0,2" "4,2,2,0" "4,1,0,4" "3,2,1,0" "2,1,0,4" "5,1,0,0"; do
  CC4_PERSIST_RUNS=$runs $B 2>/dev/null | line "pcg runs=$runs" >> $OUT
done
done
cat $OUT
