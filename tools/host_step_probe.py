#!/usr/bin/env python
"""Rate of the host-stepped surface at a large batch: CC4VecEnv.step(actions) -- upload, the step's launches, the four outputs back, one host
wait -- per step.  Usage: host_step_probe.py [envs] [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
k = int(sys.argv[2]) if len(sys.argv) > 2 else 300
for mode in (1, 0):
    env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=mode, strict=False)
    env.reset(seeds=7)
    acts = np.random.default_rng(3).integers(0, 40, size=(k, n, 5), dtype=np.int32)
    for i in range(20):
        env.step(acts[i])
    t0 = time.perf_counter()
    for i in range(20, k):
        env.step(acts[i])
    dt = time.perf_counter() - t0
    env.synchronize(); t1 = time.perf_counter()
    env.run_policy_steps(7, k, 200); env.synchronize()
    dp = (time.perf_counter() - t1) / 200
    print(f'envs {n} rng_mode {mode}: device policy + cc4_step_device {dp * 1e6:.1f} us per step = {5.0 * n / dp / 1e6:.1f} M')
    print(f'envs {n} rng_mode {mode}: CC4VecEnv.step {dt / (k - 20) * 1e6:.1f} us per step = {5.0 * n * (k - 20) / dt / 1e6:.1f} M agent-env steps/s ({env.step_kernel}, {env.launches_per_step} groups)')
    env.close()
