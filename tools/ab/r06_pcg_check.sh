#!/bin/bash
# r06: the numpy-stream kernels with the LCG window in memory (24 waves per CU): parity tests that touch them, then the rates
cd "$(dirname "$0")/../.."
line() { python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1', round(d['value']/1e6,1), 'M  wall us/step', round(d['ms_per_step']*1e3,2), ' kernel us/step', round(r['step_ms']*1e3,2), r['kernel'], 'err', d['config']['engine_error_flags'])
"; }
python -m pytest tests -m gpu -q -x -k "numpy or pcg or stream or golden or traj or scripted or persistent_kernel_and or exchange_from" 2>&1 | tail -4
for K in 20 500; do python bench.py --rng pcg64 --steps $K --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1 2>/dev/null | line "pcg64 K=$K"; done
python bench.py --rng pcg64 --steps 20 --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1 --total-envs 1024 2>/dev/null | line "pcg64 1024 envs K=20"
CC4_PERSIST_RUNS=16,1,0,0 python bench.py --steps 500 --warmup 50 --no-alt --no-cpu-baseline --min-seconds 1 2>/dev/null | line "philox K=500 runs of 16"
python bench.py --steps 500 --warmup 50 --no-alt --no-cpu-baseline --min-seconds 1 2>/dev/null | line "philox K=500 default (runs of 8)"
python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --min-seconds 1 --total-envs 5632 2>/dev/null | line "philox 5632 envs K=20"
