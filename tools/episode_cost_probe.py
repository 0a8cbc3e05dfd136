#!/usr/bin/env python
"""How much do episodes differ in step cost, how persistent is the difference, and what predicts it?  Per-episode cycle counters of the
one-wave kernel (cc4_debug_profile) over two windows of steps, against the episode's host count."""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['CC4_PHILOX_LEAN'] = '1'
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=1, strict=False)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 40, timed=False)
env.lib.cc4_debug_profile(env._h, 1, None)
out = np.zeros((n, 128), np.uint64)
tot = []
t = 40
for w in range(3):
    env.run_random_steps(1000, t, 20, timed=False); t += 20
    env.lib.cc4_debug_profile(env._h, 1, out.ctypes.data_as(ctypes.c_void_p))
    tot.append(out[:, 14].astype(np.float64).copy())
win = [tot[0], tot[1] - tot[0], tot[2] - tot[1]]          # cycles of three consecutive 20-step windows
hosts = np.array([int(env.topology(i)[27::2].sum()) for i in range(n)], np.float64)
c0, c1, c2 = (w / 20 for w in win)
print(f'{n} episodes, 20-step windows: mean cycles per step {c1.mean():.0f}, std between episodes {c1.std():.0f} ({100 * c1.std() / c1.mean():.1f} %), min {c1.min():.0f}, max {c1.max():.0f}')
print(f'correlation of an episode\'s cost in consecutive windows: {np.corrcoef(c0, c1)[0, 1]:.3f}, {np.corrcoef(c1, c2)[0, 1]:.3f}; with its host count: {np.corrcoef(hosts, c1)[0, 1]:.3f}')
for P in (256,):
    ne = n // P
    for name, key in (('e % P (as built)', None), ('snake by host count', hosts), ('snake by the previous window\'s cost', c0)):
        if key is None:
            part = np.arange(n) % P
        else:
            order = np.argsort(-key, kind='stable'); rank = np.empty(n, int); rank[order] = np.arange(n)
            row, col = rank // P, rank % P
            part = np.where(row % 2 == 0, col, P - 1 - col)
        sums = np.bincount(part, weights=c1, minlength=P)
        print(f'  partitions of {ne} episodes, {name}: partition cost max / mean = {sums.max() / sums.mean():.3f}, min / mean = {sums.min() / sums.mean():.3f}')
