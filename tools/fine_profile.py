#!/usr/bin/env python
"""Finer sections of the red policy / tick phase (a -DCC4_FINE build of libcc4.so, CC4_LIB=build_var/fine.so): red agent 0's view
of its wave, mean cycles per step.  usage: CC4_LIB=... fine_profile.py N K"""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
K = int(sys.argv[2]) if len(sys.argv) > 2 else 200
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=1)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 150, timed=False)
env.lib.cc4_debug_profile(env._h, 1, None)
ms = env.run_random_steps(1000, 150, K, timed=True)
out = np.zeros((n, 128), np.uint64)
env.lib.cc4_debug_profile(env._h, 1, out.ctypes.data_as(ctypes.c_void_p))
c = out.astype(np.float64) / K
names = ['hdr load + set_stream', 'policy (get_action + validate)', 'queue + tick', 'filter (rs_find_id)', 'store back',
         'fsm_observe', 'host choice', 'action + params']
print(f'{env.step_kernel} n={n} K={K} ms/launch {ms / K:.4f}; total cycles/step {c[:, 14].mean():.0f}; policy phase (slot 2) {c[:, 2].mean():.0f}; red agent 0 policy+tick {c[:, 16].mean():.0f}')
for k, nm in enumerate(names):
    print(f'  {nm:32s} {c[:, 104 + k].mean():8.0f}')
