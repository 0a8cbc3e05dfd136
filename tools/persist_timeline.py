import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['CC4_PERSIST_TIMELINE'] = '1'
from cage_challenge_4_amd import CC4VecEnv
env = CC4VecEnv(8192, steps=500, autoreset=True, rng_mode=1, strict=False)
env.reset(seeds=1000)
env.run_random_steps(1000, 0, 5, timed=False)
t = 5
for k in (20, 20, 20, 20, 100, 20):
    env.run_random_steps(1000, t, k, timed=True); t += k
