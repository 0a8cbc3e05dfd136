#!/usr/bin/env python
"""Cost of an autoreset launch: episodes of `steps` steps so that every `steps`-th launch regenerates all scenarios."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for mode in (1, 0):
    res = {}
    for steps in (500, 5):
        env = CC4VecEnv(n, steps=steps, autoreset=True, rng_mode=mode)
        env.reset(seeds=1000)
        env.run_random_steps(1000, 0, 50, timed=False)
        K = 300
        res[steps] = env.run_random_steps(1000, 50, K, timed=True) / K * 1e3
        env.close()
    # steps=5: done after 4 steps, so 1 launch in 5 is a reset
    print(f'rng_mode={mode} n={n}: normal launch {res[500]:.1f} us; with a reset every 5th launch {res[5]:.1f} us -> reset launch ~{5 * res[5] - 4 * res[500]:.0f} us')
