#!/usr/bin/env python
"""Parity bisecting helper: step the HIP path and the CPU oracle side by side and print the first differing bytes of the
packed state with the EnvState field map.  usage: debug_parity.py [rng_mode] [red_policy] [green_policy] [n] [steps]"""
import sys, os, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cage_challenge_4_amd import CC4VecEnv
from oracle_binding import OracleVecEnv, random_actions
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rp = int(sys.argv[2]) if len(sys.argv) > 2 else 0
gp = int(sys.argv[3]) if len(sys.argv) > 3 else 0
n = int(sys.argv[4]) if len(sys.argv) > 4 else 32
T = int(sys.argv[5]) if len(sys.argv) > 5 else 160
dev = CC4VecEnv(n, steps=150, rng_mode=mode, autoreset=True, red_policy=rp, green_policy=gp)
ora = OracleVecEnv(n, steps=150, rng_mode=mode, autoreset=True, red_policy=rp, green_policy=gp)
dev.reset(seeds=31337); ora.reset(seeds=31337)
buf = ctypes.create_string_buffer(8192); ora.lib.cc4o_layout(buf, 8192)
for t in range(-1, T):
    if t >= 0:
        a = random_actions(31337, t, n)
        dev.step(a); ora.step(a)
    bad = [i for i in range(n) if not np.array_equal(dev.get_state(i), ora.get_state(i))]
    if bad:
        i = bad[0]
        A, B = dev.get_state(i), ora.get_state(i)
        off = np.nonzero(A != B)[0]
        print('step', t, 'envs', bad[:8], 'first env', i, 'offsets', off[:30].tolist(), 'dev', A[off[:12]].tolist(), 'ora', B[off[:12]].tolist())
        if os.environ.get('CC4_DBG_VERBOSE'):
            print(buf.value.decode())
            print(ora.dump(i)[-1200:])
        break
else:
    print('no mismatch')
