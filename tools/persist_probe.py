"""Parity probe of the one-launch forms of cc4_run_random_steps (k_run_philox: the multi-step four-wave kernel of small batches;
k_run_philox1 with CC4_PERSIST=1): bursts of K steps across a scenario regeneration against the oracle.  usage: persist_probe.py [n] [rng mode]"""
import sys, os, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from cage_challenge_4_amd import CC4VecEnv
from oracle_binding import OracleVecEnv, random_actions
n, steps, seed0 = (int(sys.argv[1]) if len(sys.argv) > 1 else 8192), 150, 4242
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # 1 counter mode, 0 numpy stream
dev = CC4VecEnv(n, steps=steps, rng_mode=mode, autoreset=True)
print('kernel', dev.step_kernel, 'run kernel', dev.run_kernel, 'launches per step', dev.launches_per_step, flush=True)
ora = OracleVecEnv(n, steps=steps, rng_mode=mode, autoreset=True)
assert np.array_equal(dev.reset(seeds=seed0), ora.reset_batch(seed0))
t = 0
for K in (2, 20, 137, 20):
    dev.run_random_steps(seed0, t, K, timed=True)
    for k in range(K):
        a = random_actions(seed0, t + k, n); o = ora.step_batch(a)
    t += K
    dev.synchronize(); dev._fetch()
    bad = np.nonzero((dev._obs != o[0]).any(axis=1) | (dev._rew != o[1]) | (dev._done.astype(bool) != o[2]) | (dev._err != o[3]['err']))[0]
    print('K', K, 't', t, 'bad episodes', bad.size, bad[:10], 'actions equal', np.array_equal(dev.device_actions(), a), flush=True)
ok = np.array_equal(dev.rng_state(), ora.rng_state())
nbad = sum(not np.array_equal(dev.get_state(i), ora.get_state(i)) for i in range(0, n, 7))
print('rng equal', ok, 'packed state mismatches (every 7th)', nbad)
dev.close()
