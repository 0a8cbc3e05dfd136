#!/bin/bash
# A/B of libcc4.so variants (build_var/NAME.so) over batch sizes: bash tools/ab/ab2.sh "1024 2048" NAME [NAME ...]
sizes=$1; shift
for round in 1 2; do for v in "$@"; do for n in $sizes; do
  CC4_LIB=$PWD/build_var/$v.so python bench.py --no-alt --no-cpu-baseline --total-envs $n --min-seconds 0.4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', $n, d['roofline']['kernel'], round(d['value']/1e6,1), 'M  launch_ms', round(d['roofline']['launch_ms'],4), 'err', d['config']['engine_error_flags'])
"
done; done; done
