# A/B: cc4_run_random_steps as one launch of the plain multi-step one-wave kernel (CC4_RUN1=1) against four streams of per-step launches
export CC4_LIB=$PWD/build_var/${1:-run1}.so
CC4_RUN1=1 python tools/persist_probe.py 8192 2>&1 | tail -6
for n in 8192 4096 5120 16384; do for k in 500 20; do for mode in run1 streams; do
  if [ $mode = run1 ]; then export CC4_RUN1=1; else unset CC4_RUN1; fi
  python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), d['roofline'].get('run_kernel'), 'err', d['config']['engine_error_flags'])
"
done; done; done
