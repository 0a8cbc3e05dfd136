import os, sys, time, ctypes
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
os.environ['CC4_ROLLOUT_WATCHDOG_MS'] = '300'
import numpy as np
from cage_challenge_4_amd import CC4VecEnv
n = 8192
dev = CC4VecEnv(n, steps=60, rng_mode=1, autoreset=True, strict=False)
dev.reset(seeds=5)
print('run kernel', dev.run_kernel_for(20))
lib, h = dev.lib, dev._h
for K in (1, 2, 3):
    t0 = time.time()
    rc = lib.cc4_rollout_begin(h, K)
    print('begin', rc, lib.cc4_last_error(h) if rc else '')
    for j in range(K):
        for g in range(4):
            a = lib.cc4_rollout_wait_obs(h, g, j, None); b = lib.cc4_rollout_random_policy(h, g, j, ctypes.c_uint64(5), j, None); c = lib.cc4_rollout_publish(h, g, j, None)
            if a or b or c: print('enqueue', j, g, a, b, c, lib.cc4_last_error(h))
    e = lib.cc4_rollout_end(h)
    st = (ctypes.c_int64 * 24)(); lib.cc4_debug_rollout_state(h, st); print('state', list(st))
    print('K', K, 'end rc', e, 'in', round(time.time() - t0, 3), 's', lib.cc4_last_error(h)[:80] if e else '')
