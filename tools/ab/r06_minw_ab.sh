#!/bin/bash
# r06: the per-step kernel k_step_philox1 at 6 waves per SIMD (-DCC4_LEAN_MINW=6: 80 VGPR, 1 spill; 24 waves per CU as the one-launch kernel) against the
# compiler's own 83 VGPR (5 per SIMD): per-step launches and the learner loops that are made of them.   gpurun -- bash tools/ab/r06_minw_ab.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_minw_ab.txt
: > $OUT
for rep in 1 2; do
for lib in new minw6; do
  if [ $lib = new ]; then unset CC4_LIB; else export CC4_LIB=$PWD/build_var/$lib.so; fi
  CC4_PERSIST=0 python bench.py --warmup 5 --steps 500 --no-alt --no-cpu-baseline --min-seconds 1.0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib per-step launches K=500', round(d['value']/1e6,1), 'M', d['roofline']['kernel'])" >> $OUT
  python bench.py --warmup 5 --steps 20 --no-cpu-baseline --min-seconds 1.0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['policy_in_loop']
        print('$lib K=20 headline', round(d['value']/1e6,1), 'M; policy_in_loop', round(p['value']/1e6,1), 'grouped', round(p['grouped']['value']/1e6,1), 'rollout', round(p['rollout']['value']/1e6,1) if 'rollout' in p and p['rollout'].get('value') else None, '; facade us', d['single_env_facade']['us_per_step'])" >> $OUT
done
done
cat $OUT
