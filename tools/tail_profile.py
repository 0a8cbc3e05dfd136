#!/usr/bin/env python
"""Per-launch tail analysis: for each step, mean and max over episodes of the in-kernel cycle total, vs the launch time."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=mode)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 50, timed=False)
env.lib.cc4_debug_profile(env._h, 1, None)
prev = np.zeros((n, 128), np.uint64)
out = np.zeros((n, 128), np.uint64)
rows = []
names = ['decode', 'phish', 'fsm', 'tick', 'shuf', 'bexec', 'gexec', 'rexec', 'reasg', 'mon', 'rsc', 'in', 'obs', 'out']
for t in range(K):
    ms = env.run_random_steps(1000, 50 + t, 1, timed=True)
    env.lib.cc4_debug_profile(env._h, 1, out.ctypes.data_as(ctypes.c_void_p))
    d = (out - prev).astype(np.float64); prev = out.copy()
    tot = d[:, 14]
    w = int(tot.argmax())
    rows.append((ms * 1e3, tot.mean(), tot.max(), np.percentile(tot, 99), names[int(d[w, :14].argmax())], d[w, :14].max()))
r = np.array([(a, b, c, p) for a, b, c, p, _, _ in rows])
print(f'mode={mode} n={n}: launch us mean {r[:,0].mean():.1f}; per-episode cycles mean {r[:,1].mean():.0f}, p99 {r[:,3].mean():.0f}, max {r[:,2].mean():.0f}')
print('cycles per us implied by (max cycles / launch us):', (r[:, 2] / r[:, 0]).mean())
from collections import Counter
print('dominant phase of the slowest episode:', Counter(x[4] for x in rows).most_common(6))
print('its size (cycles):', np.mean([x[5] for x in rows]))
print('conflict-serial fraction of launches (any env):', np.mean([1.0 if (out[:,4] >= 1000000).any() else 0.0]), 'envs with conflicts so far', int((out[:,4] >= 1000000).sum()))
# ---- per-red-agent sections (slots 16 + 8*r + k): k = 0 policy, 1 exec, 2 session check, 3 fsm transition, 4 new-obs loop,
# 5 listing merge, 6 removal check, 7 validate
env2 = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=mode)
env2.reset(seeds=1000)
env2.run_random_steps(1000, 0, 250, timed=False)
env2.lib.cc4_debug_profile(env2._h, 1, None)
prev = np.zeros((n, 128), np.uint64); out = np.zeros((n, 128), np.uint64)
sec = ['policy', 'exec', 'rsc', 'fsm.trans', 'fsm.newobs', 'fsm.listing', 'fsm.removal', 'validate']
acc = np.zeros((6, 8)); cnt = np.zeros(6); mx = np.zeros((6, 8)); worst = []
for t in range(K):
    env2.run_random_steps(1000, 250 + t, 1, timed=False)
    env2.lib.cc4_debug_profile(env2._h, 1, out.ctypes.data_as(ctypes.c_void_p))
    d = (out - prev).astype(np.float64); prev = out.copy()
    a = d[:, 16:64].reshape(n, 6, 8)
    act = a[:, :, 0] > 0
    for r in range(6):
        if act[:, r].any():
            acc[r] += a[act[:, r], r].sum(0); cnt[r] += act[:, r].sum(); mx[r] = np.maximum(mx[r], a[:, r].max(0))
    w = int(d[:, 14].argmax())
    worst.append((d[w, 14], d[w, 2], a[w, :, 0].round().tolist(), {names[i]: int(d[w, i]) for i in range(14)}))
print('per red agent (active steps only): share active, mean cycles [max] per section')
for r in range(6):
    c = max(cnt[r], 1)
    print(f'  red {r}: active {cnt[r] / (K * n):.2f} ' + ' '.join(f'{sec[k]} {acc[r, k] / c:.0f}[{mx[r, k]:.0f}]' for k in range(8)))
print('slowest episode of the last launches: total, P0-P2 phase, per-agent policy cycles')
for wv in worst[-8:]:
    print('  ', round(wv[0]), round(wv[1]), wv[2], wv[3])
