export CC4_LIB=$PWD/build_var/rinl4.so
python tools/persist_probe.py 1024 2>&1 | tail -2
for n in 1024 768 512; do for k in 500 20; do for ms in 4 5; do
  export CC4_MULTISTEP=$ms
  python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('multistep=$ms n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), d['roofline'].get('run_kernel'), 'err', d['config']['engine_error_flags'])
"
done; done; done
