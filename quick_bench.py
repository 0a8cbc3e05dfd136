import sys, time
sys.path.insert(0,'/root/repo')
from cage_challenge_4_amd import CC4VecEnv, RNG_PCG64
for n in (64, 1024, 8192):
    env = CC4VecEnv(n, steps=500, autoreset=True)
    t=time.time(); env.reset(seeds=1000); print(n,'reset s', time.time()-t, flush=True)
    env.run_random_steps(1000, 0, 20, timed=False)
    t=time.time(); ms = env.run_random_steps(1000, 20, 100, timed=True); wall=time.time()-t
    print(n, 'kernel ms/step', ms/100, 'wall ms/step', wall*10, 'env-steps/s', n*100/wall, 'err', int(env.err.any()), flush=True)
    env.close()
