#!/usr/bin/env python
"""What the launch tail costs: the same total number of episodes stepped as ONE batch (one launch per step) or as G sub-batches
on G handles (each handle has its own HIP stream; the sub-batches' launches overlap, so one drains while another fills the
chip).  usage: concurrency_probe.py TOTAL K G [G ...]   (run through gpurun)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cage_challenge_4_amd import CC4VecEnv

total = int(sys.argv[1]); K = int(sys.argv[2]); groups = [int(v) for v in sys.argv[3:]] or [1, 2, 4]
os.environ.setdefault('CC4_PHILOX_LEAN', '1')
for G in groups:
    n = total // G
    envs = []
    for g in range(G):
        e = CC4VecEnv(n, steps=500, rng_mode=1, autoreset=True, strict=False)
        e.reset(seeds=np.uint64(1000 + g * n) + np.arange(n, dtype=np.uint64))
        e.run_random_steps(1000 + g * n, 0, 50, timed=False)
        envs.append(e)
    best = 1e9
    for rep in range(3):
        bar = threading.Barrier(G + 1)
        def work(e, g):
            bar.wait()
            e.run_random_steps(1000 + g * n, 50 + rep * K, K, timed=False)
        th = [threading.Thread(target=work, args=(e, g)) for g, e in enumerate(envs)]
        for t in th: t.start()
        bar.wait(); t0 = time.perf_counter()
        for t in th: t.join()
        best = min(best, time.perf_counter() - t0)
    print(f'total {total} episodes as {G} x {n} ({envs[0].step_kernel}): {5.0 * total * K / best / 1e6:.1f} M agent-env steps/s, {best / K * 1e6:.1f} us per step of the whole batch', flush=True)
    for e in envs: e.close()
