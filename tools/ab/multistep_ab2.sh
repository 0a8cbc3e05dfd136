export CC4_PERSIST_DEBUG=
for lib in multi2 multi; do
  export CC4_LIB=$PWD/build_var/$lib.so
  python tools/persist_probe.py 1024 2>&1 | tail -6
  python tools/persist_probe.py 300 2>&1 | tail -2
  for n in 1024 512 1280 256; do for k in 500 20; do
    CC4_MULTISTEP=1 python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$lib n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), 'err', d['config']['engine_error_flags'])
"
  done; done
done
