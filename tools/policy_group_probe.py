#!/usr/bin/env python
"""(needs a library built with -DCC4_POLICY_PROBE: e.g. hipcc ... -DCC4_DEV_FAST -DCC4_POLICY_PROBE -o build_var/probe.so ..., CC4_LIB=build_var/probe.so)
DESIGN 3.4 / VERDICT r04 #1: what the red policy phase costs with the agents of G episodes side by side on one wave (cc4_debug_policy_probe), on a
live 8192-episode batch at several points of its episodes.  Usage: policy_group_probe.py [envs]"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['CC4_PHILOX_LEAN'] = '1'
from cage_challenge_4_amd import CC4VecEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
env = CC4VecEnv(n, steps=500, autoreset=True, rng_mode=1, strict=False)
env.reset(seeds=1000)
out = (ctypes.c_double * 3)()
t = 0
for upto in (50, 150, 300):
    env.run_random_steps(1000, t, upto - t, timed=False); t = upto
    print(f'{n} episodes at step {t}:')
    base = None
    for G in (1, 2, 4, 8):
        rc = env.lib.cc4_debug_policy_probe(env._h, G, 20, out)
        assert rc == 0, rc
        us, cyc, waves = out[0], out[1], int(out[2])
        base = base or us
        print(f'  G = {G}: {waves} waves of {6 * G} agents, {us:8.1f} us per launch = {us * 1e3 / n:6.2f} ns per episode ({base / us:4.2f} x G = 1), '
              f'mean cycles of a wave in the phase {cyc:9.0f} = {cyc / G:8.0f} per episode')
