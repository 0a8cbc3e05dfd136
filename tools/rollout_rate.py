"""The rollout form of policy_in_loop (cc4_rollout_standin): agent-env steps/s at 8192 episodes.  usage: rollout_rate.py [K] [native|python]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cage_challenge_4_amd import CC4VecEnv
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
native = (sys.argv[2] == 'native') if len(sys.argv) > 2 else True
e = CC4VecEnv(8192, steps=500, rng_mode=1, autoreset=True, strict=False); e.reset(seeds=1000)
e.run_rollout(K, 'random', 1000, 0, native=native)
ts = []
t = K
for n in range(150):
    t0 = time.perf_counter()
    e.run_rollout(K, 'random', 1000, t, native=native); t += K
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) / K * 1e6
print('native' if native else 'python', 'groups', os.environ.get('CC4_ROLLOUT_GROUPS', 'default'), 'margin', os.environ.get('CC4_ROLLOUT_MARGIN', 'default'), 'K', K,
      ': us/step min %.1f median %.1f mean %.1f max %.1f; first ten' % (ts.min(), np.median(ts), ts.mean(), ts.max()), np.round(ts[:10], 1), 'every 25th', np.round(ts[::25], 1))
