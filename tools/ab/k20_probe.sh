for rep in 1 2 3; do for thr in 1 0; do
CC4_ENQ_THREADS=$thr python bench.py --no-alt --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); c=d['config']; print('thr=$thr', round(d['value']/1e6,1), 'median-region M', round(5*8192/c['ms_per_step_median_region']/1e3,1), 'spread', round(c['region_spread'],2), c['slowest_regions'], 'regions', c['timed_regions'])
"
done; done
