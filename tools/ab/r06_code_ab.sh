#!/bin/bash
# r06: the step kernels with ONE inlined copy of the blue / red action bodies (vbody), of the generator in rng_below and of blue_execute in step_blue_exec
# (vrng), and all of it (new: k_run_philox1 298 -> 196 KB of code) against the build before (base): parity, rates on one box, instruction-cache counters.
#   build_var/{base,vbody,vrng}.so: make -C <tree>/cage_challenge_4_amd/csrc OUT=...      gpurun -- bash tools/ab/r06_code_ab.sh
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/prof
OUT=gpurun_out/r06_code_ab.txt
: > $OUT
export TMPDIR=/tmp
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | grep -a "passed\|failed\|error" | tail -3 >> $OUT
B="python bench.py --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1.0"
for rep in 1 2; do
for lib in ${LIBS:-base new}; do
  if [ $lib = new ]; then unset CC4_LIB; else export CC4_LIB=$PWD/build_var/$lib.so; fi
  for K in 20 500; do $B --steps $K 2>/dev/null | line "$lib K=$K" >> $OUT; done
  $B --steps 20 --rng pcg64 2>/dev/null | line "$lib numpy stream K=20" >> $OUT
  $B --steps 20 --total-envs 1024 2>/dev/null | line "$lib 1024 envs K=20" >> $OUT
done
done
BENCH="python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05 --steps 100"
for lib in ${ICLIBS:-base new}; do
  if [ $lib = new ]; then unset CC4_LIB; else export CC4_LIB=$PWD/build_var/$lib.so; fi
  rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d gpurun_out/prof/icA -- $BENCH > /dev/null 2> gpurun_out/prof/icA.err
  rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAVES -d gpurun_out/prof/icB -- $BENCH > /dev/null 2> gpurun_out/prof/icB.err
  python tools/rocpd_summary.py counters k_run_philox1 gpurun_out/prof/r06_pmc_icache_k_run_philox1_8192env_$lib.json gpurun_out/prof/icA gpurun_out/prof/icB
  rm -rf gpurun_out/prof/icA gpurun_out/prof/icB
done
cat $OUT
