#!/usr/bin/env python
"""CC4_PERSIST_VERIFY=1: every one-launch call of cc4_run_random_steps is repeated with per-step launches on a shadow handle and compared
(hot rows, cold rows, outputs of every episode).  Usage: verify_probe.py [envs] [rng_mode] [calls]"""
import ctypes, os, sys, time
import numpy as np
os.environ['CC4_PERSIST_VERIFY'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
env = CC4VecEnv(n, steps=150, autoreset=True, rng_mode=mode, strict=False)
env.reset(seeds=4242)
t = 0
t0 = time.time()
for i in range(calls):
    k = (10, 20, 37, 13, 20, 64)[i % 6]
    env.run_random_steps(4242, t, k, timed=False)
    t += k
out = (ctypes.c_int64 * 2)()
env.lib.cc4_verify_stats(env._h, out)
print(f'envs {n} rng_mode {mode}: run kernel {env.run_kernel_for(20)}, {t} steps in {calls} calls, {out[0]} calls verified against per-step launches, {out[1]} disagreed, {time.time() - t0:.1f} s')
assert out[0] == calls and out[1] == 0
