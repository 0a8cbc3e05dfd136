#!/usr/bin/env python
"""What-if analysis of the launch tail: for every launch, how much shorter would the slowest episode be if phase X never
cost more than its mean (or its p90)?  Uses the in-kernel phase cycle counters (cc4_debug_profile)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=1)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 120, timed=False)
env.lib.cc4_debug_profile(env._h, 1, None)
names = ['decode', 'phish', 'fsm', 'tick', 'shuf', 'bexec', 'gexec', 'rexec', 'reasg', 'mon', 'rsc', 'in', 'obs', 'out']
prev = np.zeros((n, 128), np.uint64); out = np.zeros((n, 128), np.uint64)
D = []
for t in range(K):
    env.run_random_steps(1000, 120 + t, 1, timed=False)
    env.lib.cc4_debug_profile(env._h, 1, out.ctypes.data_as(ctypes.c_void_p))
    D.append((out - prev).astype(np.float64)[:, :15]); prev = out.copy()
D = np.stack(D)                      # [K, n, 15]
tot = D[:, :, 14]
mx = tot.max(1)
print(f'n={n} K={K}: mean episode {tot.mean():.0f}, mean launch max {mx.mean():.0f}, p99 {np.percentile(tot, 99):.0f}')
mean_p = D[:, :, :14].mean((0, 1)); p90 = np.percentile(D[:, :, :14].reshape(-1, 14), 90, axis=0)
print('phase      mean    p90    p99.9   gain if capped at mean   at p90')
for p, nm in enumerate(names):
    ex = np.maximum(0, D[:, :, p] - mean_p[p]); ex90 = np.maximum(0, D[:, :, p] - p90[p])
    g = (mx - (tot - ex).max(1)).mean(); g90 = (mx - (tot - ex90).max(1)).mean()
    print(f'{nm:8s} {mean_p[p]:7.0f} {p90[p]:7.0f} {np.percentile(D[:, :, p], 99.9):7.0f}   {g:7.0f}   {g90:7.0f}')
allex = np.maximum(0, D[:, :, :14] - p90[None, None, :]).sum(2)
print('all phases capped at their p90: launch max ->', (tot - allex).max(1).mean())
for combo in (['phish', 'reasg'], ['phish', 'reasg', 'bexec'], ['phish', 'reasg', 'bexec', 'rsc'], ['rexec', 'fsm']):
    ex = sum(np.maximum(0, D[:, :, names.index(c)] - p90[names.index(c)]) for c in combo)
    print('capped at p90:', combo, '->', (tot - ex).max(1).mean())
