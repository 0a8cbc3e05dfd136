import sys, os, time, statistics
sys.path.insert(0, '/root/repo')
import numpy as np
from cage_challenge_4_amd import CC4VecEnv
n=8192; K=20
e = CC4VecEnv(n, steps=500, rng_mode=1, autoreset=True, strict=False)
e.reset(seeds=np.uint64(1000) + np.arange(n, dtype=np.uint64))
e.run_random_steps(1000, 0, 50, timed=False); e.synchronize()
t=50
a=[];b=[];c=[];ms=[]
for i in range(200):
    e.synchronize(); t0=time.perf_counter()
    m=e.run_random_steps(1000, t, K, timed=True); t1=time.perf_counter()
    e.synchronize(); t2=time.perf_counter(); t+=K
    a.append(t1-t0); b.append(t2-t1); ms.append(m)
print('run_random_steps %.1f us, trailing synchronize %.1f us, on-stream %.1f us' % (statistics.mean(a)*1e6, statistics.mean(b)*1e6, statistics.mean(ms)*1e3))
os.environ['CC4_HOST_PROF']='1'
for i in range(3):
    e.run_random_steps(1000, t, K, timed=True); t+=K
e.close()
