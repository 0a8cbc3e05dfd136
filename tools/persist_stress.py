"""Stress of the persistent kernel's shared tail (items of a partition taken by other CUs of the XCD behind an agent-scope acquire): many short
calls -- every call ends in a tail -- of varying length against the oracle, several batch sizes.  usage: persist_stress.py [rounds] [rng mode]"""
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cage_challenge_4_amd import CC4VecEnv
from oracle_binding import OracleVecEnv, random_actions
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1 counter mode (k_run_philox1), 0 numpy stream (k_run_pcg)
bad_total = 0
for rnd in range(rounds):
    for n in (8192, 6400, 7001, 12288):
        steps, seed0 = 120, 9000 + 17 * rnd + n
        dev = CC4VecEnv(n, steps=steps, rng_mode=mode, autoreset=True)
        assert dev.run_kernel == ('k_run_pcg', 'k_run_philox1')[mode], dev.run_kernel
        ora = OracleVecEnv(n, steps=steps, rng_mode=mode, autoreset=True)
        assert np.array_equal(dev.reset(seeds=seed0), ora.reset_batch(seed0))
        t = 0
        for K in (10, 11, 13, 10, 25, 10, 12, 40, 10, 10, 15, 10, 33, 10, 10, 21, 10):
            dev.run_random_steps(seed0, t, K, timed=False)
            for k in range(K):
                o = ora.step_batch(random_actions(seed0, t + k, n))
            t += K
            dev.synchronize(); dev._fetch()
            bad = np.nonzero((dev._obs != o[0]).any(axis=1) | (dev._rew != o[1]) | (dev._done.astype(bool) != o[2]) | (dev._err != o[3]['err']))[0]
            if bad.size:
                print('MISMATCH n', n, 'round', rnd, 'K', K, 't', t, bad[:10], flush=True); bad_total += 1; break
        ok = np.array_equal(dev.rng_state(), ora.rng_state())
        nbad = sum(not np.array_equal(dev.get_state(i), ora.get_state(i)) for i in range(0, n, 5))
        print('n', n, 'round', rnd, 'steps', t, 'rng equal', ok, 'state mismatches (every 5th)', nbad, flush=True)
        bad_total += (not ok) + nbad
        dev.close(); ora.close()
print('persist_stress: problems =', bad_total)
