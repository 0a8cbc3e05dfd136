#!/usr/bin/env python
"""Host-side composition of a short timed region: wall time of cc4_run_random_steps(k) against the on-stream time of its k steps
(first kernel start -> last kernel end on the slowest stream), for k = 1, 2, 5, 20; then three calls under CC4_HOST_PROF.
usage: region_host_probe.py [N]   (through gpurun; CC4_GROUPS=1 for one launch per step)"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
e = CC4VecEnv(n, steps=500, rng_mode=1, autoreset=True, strict=False)
e.reset(seeds=np.uint64(1000) + np.arange(n, dtype=np.uint64))
e.run_random_steps(1000, 0, 50, timed=False); e.synchronize()
t = 50
for K in (1, 2, 5, 20):
    a, ms = [], []
    for i in range(200):
        e.synchronize(); t0 = time.perf_counter()
        m = e.run_random_steps(1000, t, K, timed=True); t1 = time.perf_counter()
        t += K
        a.append(t1 - t0); ms.append(m)
    print('n=%d launches/step %d K=%2d: call %.1f us, on-stream %.1f us, outside the kernels %.1f us' % (n, e.launches_per_step if hasattr(e, 'launches_per_step') else -1, K, statistics.median(a) * 1e6, statistics.median(ms) * 1e3, (statistics.median(a) - statistics.median(ms) * 1e-3) * 1e6))
os.environ['CC4_HOST_PROF'] = '1'
for i in range(2):
    e.run_random_steps(1000, t, 20, timed=True); t += 20
e.close()
