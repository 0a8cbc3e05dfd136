#!/bin/bash
# One-rank communicator (CC4_BENCH_FORCE_DIST=1), 1024 / 8192 episodes, 1..4 launches per step: rate, host cost per step and whether
# the groups' kernels overlap on the chip (tools/stream_overlap.py on a rocprofv3 kernel trace).  usage: through gpurun.
OUT=gpurun_out/prof; mkdir -p $OUT; export TMPDIR=/tmp
R=$OUT/${1:-r03}_exchange_overlap.txt; : > $R
for N in 1024 8192; do for G in 1 2 3 4; do
  echo "== n=$N groups=$G, exchange on a one-rank communicator" >> $R
  CC4_BENCH_FORCE_DIST=1 CC4_GROUPS=$G CC4_HOST_PROF=1 timeout 200 python bench.py --no-alt --no-cpu-baseline --min-seconds 0.15 --total-envs $N 2> $OUT/xo.err | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['config']['per_rank'][0]
        print('  %.1f M agent-env steps/s, %.1f us/step, launch_ms %.1f us, host launch %.1f us/step, all-gather enqueue %.1f us/step, waited %s of %s' % (d['value']/1e6, d['ms_per_step']*1e3, d['roofline'].get('launch_ms', d['roofline'].get('kernel_ms', -1e-3))*1e3, p['host_launch_us_per_step'], p['host_allgather_enqueue_us_per_step'], p['steps_that_waited_for_an_allgather'], p['allgathers_issued']))
" >> $R 2>&1
  rm -rf $OUT/xo_tr
  CC4_BENCH_FORCE_DIST=1 CC4_GROUPS=$G timeout 300 rocprofv3 --kernel-trace -d $OUT/xo_tr -- python bench.py --no-alt --no-cpu-baseline --min-seconds 0.05 --total-envs $N > /dev/null 2> $OUT/xo_tr.err
  python tools/stream_overlap.py $OUT/xo_tr 2>&1 | sed 's/^/  /' >> $R
done; done
rm -rf $OUT/xo_tr
cat $R
