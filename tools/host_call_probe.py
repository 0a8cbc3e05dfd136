"""Where the wall clock of a 20-step timed region goes on the host: cc4_run_random_steps (launch .. synchronised), the bench's extra synchronize(), Python."""
import os, sys, time, statistics as st
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cage_challenge_4_amd import CC4VecEnv
e = CC4VecEnv(8192, steps=500, rng_mode=1, autoreset=True, strict=False); e.reset(seeds=1000)
e.run_random_steps(1000, 0, 20, timed=True)
t = 20
call, sync, ms, empty = [], [], [], []
for i in range(300):
    e.synchronize()
    t0 = time.perf_counter()
    m = e.run_random_steps(1000, t, 20, timed=True)
    t1 = time.perf_counter()
    e.synchronize()
    t2 = time.perf_counter()
    e.synchronize()
    t3 = time.perf_counter()
    t += 20
    call.append(t1 - t0); sync.append(t2 - t1); ms.append(m * 1e-3); empty.append(t3 - t2)
us = lambda v: round(st.median(v) * 1e6, 1)
print('20-step call: wall', us(call), 'us; kernel (HIP events)', us(ms), 'us; outside the kernel', round(us(call) - us(ms), 1), 'us; the synchronize() behind it', us(sync), 'us; a second synchronize()', us(empty), 'us')
