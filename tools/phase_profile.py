#!/usr/bin/env python
"""Per-phase cycle breakdown of k_step (lane-0 walk) from the in-kernel counters (cc4_debug_profile)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 0
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=mode)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 50, timed=False)
env.lib.cc4_debug_profile(env._h, 1, None)
ms = env.run_random_steps(1000, 50, K, timed=True)
out = np.zeros((n, 128), np.uint64)
env.lib.cc4_debug_profile(env._h, 1, out.ctypes.data_as(ctypes.c_void_p))
names = ['blue decode/queue', 'green policy draws', 'red FSM policy', 'queue tick', 'shuffle', 'blue exec', 'green exec', 'red exec',
         'reassign', 'monitor x5', 'red session check', 'stage in', 'flat obs', 'stage out', 'TOTAL', '-']
c = out.astype(np.float64) / K
tot = c[:, 14].mean()
print(f'rng_mode={mode} n={n} K={K} kernel ms/launch {ms / K:.4f}; mean cycles/step per episode {tot:.0f}; max over episodes {c[:, 14].max():.0f}')
for i in list(range(14)):
    print(f'{names[i]:22s} {c[:, i].mean():10.0f} cyc  {100 * c[:, i].mean() / tot:5.1f}%   (max {c[:, i].max():.0f})')
ty = ['DRS', 'Aggressive', 'Stealth', 'Deception', 'Exploit', 'PrivEsc', 'Impact', 'Degrade', 'Withdraw', 'Sleep', 'Invalid', 'None', '12', '13', '14', '15']
tt = out[:, 64:96].astype(np.float64).reshape(n, 16, 2).sum(0)
if tt[:, 1].sum() > 0:
    print('red action execution (own-wave, non-conflicting): type, count per episode-step, mean cycles')
    for i in range(16):
        if tt[i, 1] > 0:
            print(f'  {ty[i]:10s} {tt[i, 1] / (n * K):6.3f}  {tt[i, 0] / tt[i, 1]:8.0f}')
bt = out[:, 108:124].astype(np.float64).reshape(n, 8, 2).sum(0)
if bt[:, 1].sum() > 0:
    print('blue action execution (four-wave kernel, independent case): type, count per episode-step, mean cycles')
    for i, nm in enumerate(['Sleep', 'Monitor', 'Analyse', 'Remove', 'Restore', 'DeployDecoy', 'Block', 'Allow']):
        if bt[i, 1] > 0:
            print(f'  {nm:11s} {bt[i, 1] / (n * K):6.3f}  {bt[i, 0] / bt[i, 1]:8.0f}')
print('green action waves (AccessService list, LocalWork list): mean cycles', (out[:, 96].mean() / K).round(), (out[:, 97].mean() / K).round())
