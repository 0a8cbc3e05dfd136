#!/bin/bash
# r06: XCD pools (RunArgs.pool) against the per-CU partitions of r05, one box, one library (build_var/pool.so = -DCC4_DEV_FAST build of the tree).
#   gpurun -- bash tools/ab/r06_pool_ab.sh
cd "$(dirname "$0")/../.."
export CC4_LIB=$PWD/build_var/pool.so
mkdir -p gpurun_out
OUT=gpurun_out/r06_pool_ab.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'frac', round(r['frac'],3), 'err', d['config']['engine_error_flags'])
"; }
for rep in 1 2; do
for K in 20 500; do
  for pool in 0 1; do
    CC4_PERSIST_POOL=$pool python bench.py --steps $K --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0 2>/dev/null | line "pool=$pool K=$K" >> $OUT
  done
done
CC4_PERSIST_POOL=0 CC4_PERSIST_ORDER=3 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0 2>/dev/null | line "pool=0 order=3 (L1 invalidate per item) K=20" >> $OUT
done
for n in 16384 32768 5632; do for pool in 0 1; do
  CC4_PERSIST_POOL=$pool python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --min-seconds 0.6 --total-envs $n 2>/dev/null | line "pool=$pool K=20 envs=$n" >> $OUT
done; done
echo "## timelines" >> $OUT
CC4_PERSIST_POOL=1 python tools/persist_timeline.py 2>&1 | grep "cc4 timeline" >> $OUT
echo "## self-check (CC4_PERSIST_VERIFY) in pool mode" >> $OUT
for n in 8192 5632 16384 7001; do CC4_PERSIST_POOL=1 timeout 600 python tools/verify_probe.py $n 1 30 2>&1 | tail -1 >> $OUT; done
echo "## stress against the oracle, pool mode" >> $OUT
CC4_PERSIST_POOL=1 timeout 900 python tools/persist_stress.py 1 1 2>&1 | tail -6 >> $OUT
cat $OUT
