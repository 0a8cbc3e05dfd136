#!/usr/bin/env python
"""What a short timed region costs (the driver runs bench.py --steps 20 --warmup 5: every 20 steps are bracketed by a
synchronisation of all streams).  Regions of K steps, with and without the HIP-event timing inside cc4_run_random_steps.
usage: short_region_probe.py N K [K ...]   (through gpurun)"""
import sys, os, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cage_challenge_4_amd import CC4VecEnv

n = int(sys.argv[1]); Ks = [int(v) for v in sys.argv[2:]] or [20, 100, 500]
e = CC4VecEnv(n, steps=500, rng_mode=1, autoreset=True, strict=False)
e.reset(seeds=np.uint64(1000) + np.arange(n, dtype=np.uint64))
e.run_random_steps(1000, 0, 50, timed=False); e.synchronize()
t = 50
for K in Ks:
    for timed in (True, False, True, False):
        secs = []
        reps = max(10, int(0.3 / (K * 60e-6)))
        for _ in range(reps):
            e.synchronize(); t0 = time.perf_counter()
            e.run_random_steps(1000, t, K, timed=timed)
            e.synchronize(); secs.append(time.perf_counter() - t0); t += K
        med = statistics.median(secs)
        print(f'n={n} K={K:4d} events={"yes" if timed else "no "}: mean {sum(secs) / len(secs) / K * 1e6:6.2f} us/step, median region {med / K * 1e6:6.2f} us/step, '
              f'region overhead vs 500-step rate: see below; {5.0 * n * K * len(secs) / sum(secs) / 1e6:.1f} M', flush=True)
e.close()
