# the multi-step kernels (k_run_philox: five blocks per CU; k_run_philox8: eight) against the per-step launches, by batch size
export CC4_LIB=$PWD/build_var/${1:-rinl8}.so
python tools/persist_probe.py 1024 2>&1 | tail -2
python tools/persist_probe.py 2048 2>&1 | tail -2
for n in 1024 1280 1536 2048; do for k in 500 20; do for ms in auto 0 8; do
  if [ $ms = auto ]; then unset CC4_MULTISTEP; else export CC4_MULTISTEP=$ms; fi
  python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('multistep=$ms n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), d['roofline'].get('run_kernel'), 'err', d['config']['engine_error_flags'])
"
done; done; done
