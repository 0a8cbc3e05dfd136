# A/B: -mllvm -disable-machine-licm (nomlicm) against the round's flags (ref): per-step kernel at 8192 / 16384, the multi-step kernels at their sizes, the persistent kernel
for lib in ref nomlicm; do
  export CC4_LIB=$PWD/build_var/$lib.so
  [ $lib = nomlicm ] && { python tools/persist_probe.py 1024 2>&1 | tail -1; python tools/persist_probe.py 4096 2>&1 | tail -1; python tools/persist_probe.py 8192 2>&1 | tail -1; }
  for cfg in "8192 500" "8192 20" "16384 500" "1024 500" "1024 20" "2048 500" "4096 500" "5120 500"; do set -- $cfg
    python bench.py --no-alt --no-cpu-baseline --steps $2 --warmup 5 --total-envs $1 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib n=$1 K=$2', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), d['roofline'].get('run_kernel'), 'err', d['config']['engine_error_flags'])
"
  done
  CC4_PERSIST=1 CC4_RUN1=0 python bench.py --no-alt --no-cpu-baseline --steps 500 --warmup 5 --total-envs 8192 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib persist n=8192 K=500', round(d['value']/1e6,1), 'M', d['roofline'].get('run_kernel'))
"
done
