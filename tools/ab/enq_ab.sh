# A/B of the enqueue threads (CC4_ENQ_THREADS) on one box: parity probe, then bench at K = 20 / 500, 8192 and 4096 episodes
export CC4_LIB=$PWD/build_var/enq.so
python tools/persist_probe.py 8192 2>&1 | tail -6
for rep in 1 2; do for mode in off thr thrjoin; do for n in 8192 4096; do for k in 20; do
  case $mode in off) export CC4_ENQ_THREADS=0; unset CC4_ENQ_JOIN;; thr) export CC4_ENQ_THREADS=1; unset CC4_ENQ_JOIN;; thrjoin) export CC4_ENQ_THREADS=1; export CC4_ENQ_JOIN=1;; esac
  python bench.py --no-alt --no-cpu-baseline --steps $k --warmup 5 --total-envs $n 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$mode n=$n K=$k', round(d['value']/1e6,1), 'M  ms_per_step', round(d['ms_per_step'],5), 'launch_ms', round(d['roofline']['launch_ms'],5), 'err', d['config']['engine_error_flags'])
"
done; done; done; done
for mode in off thr thrjoin; do
  case $mode in off) export CC4_ENQ_THREADS=0; unset CC4_ENQ_JOIN;; thr) export CC4_ENQ_THREADS=1; unset CC4_ENQ_JOIN;; thrjoin) export CC4_ENQ_THREADS=1; export CC4_ENQ_JOIN=1;; esac
  echo $mode; python tools/region_host_probe.py 8192 2>&1 | grep "^n="
done
