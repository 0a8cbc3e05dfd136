#!/bin/bash
# Exchange on a one-rank communicator (CC4_BENCH_FORCE_DIST=1): how the number of episode groups and the way they came about
# (library's choice + regroup at cc4_comm_init, CC4_GROUPS, CC4_EXCHANGE_GROUPS) change the rate.  usage: through gpurun.
N=${1:-1024}
B="python bench.py --no-alt --no-cpu-baseline --total-envs $N --min-seconds 0.15"
r() { "$@" 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); p=d['config']['per_rank'][0]; print('   %.1f M  %.1f us/step  lps %d host launch %.1f ag %.1f waited %s/%s' % (d['value']/1e6, d['ms_per_step']*1e3, p['launches_per_step'], p['host_launch_us_per_step'], p['host_allgather_enqueue_us_per_step'], p['steps_that_waited_for_an_allgather'], p['allgathers_issued']))
"; }
export CC4_BENCH_FORCE_DIST=1
echo "n=$N"
echo "auto (regroup at comm_init)"; r $B
for G in 1 2 3 4; do echo "auto, CC4_EXCHANGE_GROUPS=$G"; CC4_EXCHANGE_GROUPS=$G r $B; done
echo "CC4_GROUPS=1"; CC4_GROUPS=1 r $B
echo "CC4_GROUPS=1 + four streams exist"; CC4_GROUPS=1 CC4_EXP_PROBE=1 r $B
echo "CC4_GROUPS=1 + four streams exist + probe"; CC4_GROUPS=1 CC4_EXP_PROBE=2 r $B
echo "CC4_GROUPS=2 + four streams exist + probe"; CC4_GROUPS=2 CC4_EXP_PROBE=2 r $B
