#!/bin/bash
# r06: compiler flags again, now that the step kernels are bound by vector issue slots (r04's six rounds were run on latency-bound kernels):
#   new = the Makefile's flags; fo3 = -O3 instead of -O2; flicm = without -disable-machine-licm; fstruct = without the structurizecfg pair;
#   funroll = without the two unroll thresholds.     gpurun -- bash tools/ab/r06_flags_ab.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_flags_ab.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0"
for rep in 1 2; do
for lib in new fo3 flicm fstruct funroll; do
  if [ $lib = new ]; then unset CC4_LIB; else export CC4_LIB=$PWD/build_var/$lib.so; fi
  $B --steps 500 2>/dev/null | line "$lib K=500" >> $OUT
  $B --steps 20 --total-envs 1024 2>/dev/null | line "$lib 1024 envs K=20" >> $OUT
done
done
cat $OUT
