#!/usr/bin/env python
"""Soak of the exchange from inside the one-launch kernels (not collected by pytest; run on the GPU box: python tests/soak_exchange.py [envs] [calls] [rng mode]).
Every step's gathered rows of every call against the oracle, calls of 10..90 steps (longer than the ring of 32 every other time), the exchange slowed down by a
random 0..60 us per all-gather, across regenerations; at the end the generator position and a sample of the packed states."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
from oracle_binding import OracleVecEnv, random_actions
from test_hip_parity import _pack, _one_rank_comm
from cage_challenge_4_amd import CC4VecEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 60
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
seed0, steps = 2468, 120
dev = CC4VecEnv(n, steps=steps, rng_mode=mode, autoreset=True)
dev.reset(seeds=seed0)
_one_rank_comm(dev)
ora = OracleVecEnv(n, steps=steps, rng_mode=mode, autoreset=True); ora.reset_batch(seed0)
ora.lib.cc4o_set_threads(min(16, os.cpu_count() or 1))      # (a cgroup-limited box: more OpenMP threads than cores it may use only spin)
rng = np.random.default_rng(99)
t = bad_total = 0
for c in range(calls):
    K = int(rng.integers(10, 91))
    dev._chk(dev.lib.cc4_debug_comm_delay_us(dev._h, int(rng.integers(0, 61))), 'cc4_debug_comm_delay_us')
    dev.gather_log(K)
    dev.run_random_steps(seed0, t, K, timed=False)
    got = dev.get_gather_log(1, 0, K)
    for k in range(K):
        o = ora.step_batch(random_actions(seed0, t + k, n))
        bad = np.nonzero((got[k] != _pack(o[0].astype(np.uint8))).any(axis=1))[0]
        if bad.size:
            bad_total += 1
            print(f'call {c}, step {t + k} (K = {K}): {bad.size} rows differ, first {bad[:5].tolist()}')
    t += K
xi = dev.exchange_info()
ok_rng = np.array_equal(dev.rng_state(), ora.rng_state())
ok_state = all(np.array_equal(dev.get_state(i), ora.get_state(i)) for i in range(0, n, max(1, n // 64)))
print(f'envs {n} rng_mode {mode}: {calls} calls, {t} steps, run kernel {dev.run_kernel_for(20)}, exchange {xi}; steps with wrong gathered rows {bad_total}; '
      f'generator position equal {ok_rng}, sampled states equal {ok_state}')
assert bad_total == 0 and ok_rng and ok_state and xi['watchdog_timeouts'] == 0
