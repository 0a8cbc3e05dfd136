#!/usr/bin/env python
"""Per-wave arrival time at the barrier that ends the policy phase (debug slots 100..103 of cc4_debug_profile)."""
import ctypes, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n, K = 1024, 300
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=1)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 100, timed=False)
env.lib.cc4_debug_profile(env._h, 1, None)
env.run_random_steps(1000, 100, K, timed=False)
out = np.zeros((n, 128), np.uint64)
env.lib.cc4_debug_profile(env._h, 1, out.ctypes.data_as(ctypes.c_void_p))
w = out[:, 100:104].astype(np.float64) / K
print('mean cycles from kernel start to the end of the policy phase, per wave:', w.mean(0).round(0), ' max wave mean', w.max(1).mean().round(0))
