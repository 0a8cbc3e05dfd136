#!/usr/bin/env python
"""numpy-stream kernel, lane-parallel green actions (wave_green_exec): batches per step, why they ended, cycles per part."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=0)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 50, timed=False)
env.lib.cc4_debug_profile(env._h, 1, None)
prev = np.zeros((n, 128), np.uint64); out = np.zeros((n, 128), np.uint64)
names = ['batches', 'phish', 'walk_hard', 'collision', 'cyc_walk', 'cyc_commit', 'cyc_serial', 'rounds']
tot = np.zeros(8); worst = []
for t in range(K):
    env.run_random_steps(1000, 50 + t, 1, timed=False)
    env.lib.cc4_debug_profile(env._h, 1, out.ctypes.data_as(ctypes.c_void_p))
    d = (out - prev).astype(np.float64); prev = out.copy()
    tot += d[:, 64:72].mean(axis=0)
    w = int(d[:, 14].argmax())
    worst.append((d[w, 14], d[w, 6], *d[w, 64:72]))
print('mean per episode-step:', dict(zip(names, np.round(tot / K, 2))))
print('slowest episode of each launch: total, gexec,', names)
for r in worst[:20]: print('  ', [int(v) for v in r])
