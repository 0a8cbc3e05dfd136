#!/usr/bin/env python
"""Race hunt: repeated full-batch parity runs (HIP path vs CPU oracle, every step, every episode) with different seeds,
policies and RNG modes.  usage: stress_parity.py [rounds] [envs] [steps] [episode_steps] [seed_base]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from cage_challenge_4_amd import CC4VecEnv
from oracle_binding import OracleVecEnv, random_actions
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
T = int(sys.argv[3]) if len(sys.argv) > 3 else 220
EP = int(sys.argv[4]) if len(sys.argv) > 4 else 150   # episode length (autoreset after it)
SEED_BASE = int(sys.argv[5]) if len(sys.argv) > 5 else 50000
bad = 0
for k in range(rounds):
    # (rng mode, red policy, counter-mode kernel: 0 = four wavefronts per episode, 1 = one)
    # ... and, last field, the built-in blue policy (acts for the agents whose action index is negative: every third here)
    for mode, rp, lean, bp in ((1, 0, 0, 0), (1, 0, 1, 0), (1, 3, 1, 1), (0, 0, 0, 0), (1, 2, 0, 1), (1, 2, 1, 0), (0, 3, 0, 1), (0, 2, 0, 0)):
        seed = SEED_BASE + 1000 * k + 17 * mode + rp
        os.environ['CC4_PHILOX_LEAN'] = str(lean)
        dev = CC4VecEnv(n, steps=EP, rng_mode=mode, autoreset=True, red_policy=rp, blue_policy=bp)
        ora = OracleVecEnv(n, steps=EP, rng_mode=mode, autoreset=True, red_policy=rp, blue_policy=bp)
        assert np.array_equal(dev.reset(seeds=seed), ora.reset_batch(seed))
        ok = True
        for t in range(T):
            a = random_actions(seed, t, n)
            if bp:
                a[(np.arange(n)[:, None] + np.arange(5)[None, :] + t) % 3 == 0] = -1
            d = dev.step(a); o = ora.step_batch(a)      # OpenMP over episodes: large batches in seconds
            if not (np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1]) and np.array_equal(d[2], o[2])):
                print('MISMATCH round', k, 'mode', mode, 'policy', rp, 'kernel', dev.step_kernel, 'step', t, flush=True); ok = False; bad += 1; break
        if ok:
            for i in range(0, n, 3):
                if not np.array_equal(dev.get_state(i), ora.get_state(i)):
                    print('STATE MISMATCH round', k, 'mode', mode, 'policy', rp, 'env', i, flush=True); bad += 1; break
        dev.close(); ora.close()
    print('round', k, 'done', flush=True)
print('stress_parity: mismatches =', bad)
