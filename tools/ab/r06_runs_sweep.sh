#!/bin/bash
# r06: run-length patterns of a 20-step call (CC4_PERSIST_RUNS="SA,SB,nB,single": nB runs of SB and `single` single steps close the call, runs of SA fill the rest) and the helper threshold
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_runs_sweep.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), 'err', d['config']['engine_error_flags'])
"; }
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0 --steps 20"
for rep in 1 2; do
for runs in "4,1,0,0" "3,4,2,0" "5,1,0,0" "2,4,4,0" "3,1,0,2" "2,1,0,0" "4,1,0,4"; do
  CC4_PERSIST_RUNS=$runs $B 2>/dev/null | line "runs=$runs" >> $OUT
done
for thr in 2 4 8 32 1000; do CC4_PERSIST_THR=$thr $B 2>/dev/null | line "thr=$thr (runs 4,1,0,0)" >> $OUT; done
done
for K in 10 12 16 24 32 40 50 64 100; do python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 0.7 --steps $K 2>/dev/null | line "K=$K default runs" >> $OUT; done
cat $OUT
