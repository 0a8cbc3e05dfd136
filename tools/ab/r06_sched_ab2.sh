#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_sched_ab2.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
export CC4_LIB=$PWD/build_var/bal6.so
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0"
for rep in 1 2; do
  for K in 20 500; do
    CC4_PERSIST_SCHED=0 $B --steps $K 2>/dev/null | line "sched=0 K=$K" >> $OUT
    CC4_PERSIST_SCHED=2 $B --steps $K 2>/dev/null | line "sched=2 thr=8 K=$K" >> $OUT
  done
  for thr in 0 2 4 16 1000; do CC4_PERSIST_SCHED=2 CC4_PERSIST_THR=$thr $B --steps 20 2>/dev/null | line "sched=2 thr=$thr K=20" >> $OUT; done
  for runs in "4,2,1,1" "3,1,0,2" "4,1,0,0" "4,2,2,0" "5,2,1,1" "3,2,1,0"; do CC4_PERSIST_SCHED=2 CC4_PERSIST_RUNS=$runs $B --steps 20 2>/dev/null | line "sched=2 runs=$runs K=20" >> $OUT; done
done
CC4_PERSIST_SCHED=2 python tools/persist_timeline.py 2>&1 | grep "cc4 timeline" | grep -v "XCD [1-7]" >> $OUT
echo "## self-check (CC4_PERSIST_VERIFY), schedule 2" >> $OUT
for n in 6500 8192 16384; do CC4_PERSIST_SCHED=2 timeout 600 python tools/verify_probe.py $n 1 60 2>&1 | tail -1 >> $OUT; done
cat $OUT
