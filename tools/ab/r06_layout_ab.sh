#!/bin/bash
# r06: (1) the 5952-byte agent part (5 LDS granules) at 5 vs 6 waves per SIMD, per-CU partitions (pool=0); (2) XCD pools with the episodes' rows in
# fine-grained / uncached device memory (does the L1 keep their lines?  tools/micro/l1_inv_scope); one box.   gpurun -- bash tools/ab/r06_layout_ab.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out
OUT=gpurun_out/r06_layout_ab.txt
: > $OUT
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
for m in 0 1 2; do ./tools/micro/l1_inv_scope $m >> $OUT; done
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0"
for rep in 1 2; do
for lib in lay5 lay6; do
  export CC4_LIB=$PWD/build_var/$lib.so
  for K in 20 500; do
    CC4_PERSIST_POOL=0 CC4_PERSIST_DEBUG=1 $B --steps $K 2>gpurun_out/dbg_$lib.txt | line "$lib pool=0 K=$K" >> $OUT
  done
  CC4_PERSIST=0 $B --steps 500 2>/dev/null | line "$lib per-step launches K=500" >> $OUT
done
done
grep "persistent kernel" gpurun_out/dbg_lay5.txt | head -3 >> $OUT; grep "persistent kernel" gpurun_out/dbg_lay6.txt | head -3 >> $OUT
export CC4_LIB=$PWD/build_var/lay5.so
for mem in 1 2; do
  for pool in 0 1; do
  CC4_EXP_MEM=$mem CC4_PERSIST_POOL=$pool $B --steps 20 2>/dev/null | line "lay5 mem=$mem pool=$pool K=20" >> $OUT
  done
  echo "## self-check, pool mode, CC4_EXP_MEM=$mem" >> $OUT
  for n in 5632 8192; do CC4_EXP_MEM=$mem CC4_PERSIST_POOL=1 timeout 600 python tools/verify_probe.py $n 1 60 2>&1 | tail -1 >> $OUT; done
done
cat $OUT
