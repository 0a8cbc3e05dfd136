export CC4_LIB=$PWD/$1
export CC4_RUN1=0
for k in 10 20 32 50 100 500; do for mode in persist streams; do
  if [ $mode = persist ]; then unset CC4_PERSIST; else export CC4_PERSIST=0; fi
  python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs 8192 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$mode K=$k', round(d['value']/1e6,1), 'M  wall us/region', round(d['ms_per_step']*1e3*$k,1), ' kernel us/region', round(r['launch_ms']*1e3*$k,1), ' per step', round(d['ms_per_step']*1e3,2), round(r['launch_ms']*1e3,2))
"
done; done
