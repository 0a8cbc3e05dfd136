"""debug: the stepping of a rollout alone -- every pass pre-published (CC4_ROLLOUT_PREPUBLISH=1), nothing on the policy stream."""
import os, sys, time
os.environ['CC4_ROLLOUT_PREPUBLISH'] = '1'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cage_challenge_4_amd import CC4VecEnv
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
e = CC4VecEnv(8192, steps=500, rng_mode=1, autoreset=True, strict=False); e.reset(seeds=1000)
lib, h = e.lib, e._h
ts = []
for n in range(100):
    t0 = time.perf_counter()
    assert lib.cc4_rollout_begin(h, K) == 0
    rc = lib.cc4_rollout_end(h)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts[5:]) / K * 1e6
print('free-running rollout, groups', os.environ.get('CC4_ROLLOUT_GROUPS', 'default'), 'margin', os.environ.get('CC4_ROLLOUT_MARGIN', 'default'), 'K', K, ': us/step median %.1f min %.1f' % (np.median(ts), ts.min()), 'rc', rc)
