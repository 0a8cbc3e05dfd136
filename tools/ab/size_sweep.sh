# kernel choice by batch size (cc4_create's table) re-measured: bench.py at K = 500 for the four-wave kernel with register budget 1 / 7 / 8 and the
# one-wave kernel, per size
for n in 1280 1536 2048 3072 4096; do
  for cfg in "lean=0 minw=1" "lean=0 minw=7" "lean=0 minw=8" "lean=1 minw=1" "auto"; do
    if [ "$cfg" = auto ]; then unset CC4_PHILOX_LEAN CC4_PHILOX_MINW; else
      export CC4_PHILOX_LEAN=$(echo $cfg | sed 's/lean=\([01]\).*/\1/'); export CC4_PHILOX_MINW=$(echo $cfg | sed 's/.*minw=//'); fi
    CC4_MULTISTEP=0 python bench.py --no-alt --no-cpu-baseline --total-envs $n --min-seconds 0.4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('n=$n $cfg:', round(d['value']/1e6,1), 'M  launch_ms', round(d['roofline']['launch_ms'],4), d['roofline']['kernel'], 'launches', d['roofline']['launches_per_step'], 'err', d['config']['engine_error_flags'])
"
  done
done
