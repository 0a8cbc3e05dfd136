"""Golden trajectories of the COUNTER (Philox) mode INCLUDING ITS SCENARIO GENERATION, recorded from the REAL reference (CybORG v4 at
/root/reference) built, reset and stepped under oracle/refgen/philox_proxy.PhiloxProxy armed for generation (compare_ctrgen.Pair):
scenario #1 at construction, #2 at the wrapper's reset(), `steps` steps of seeded random blue actions, scenario #3 (reset on the
running key: what the kernels' in-kernel autoreset generates), `more` further steps.  Data only: key, actions, and the reference's
outputs -- flat observations, action masks, rewards, dones, and a digest of the full canonical state dump after every reset.

usage: python make_ctrgen_golden.py      # writes tests/golden/ctrgen_*.npz"""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from compare import RED, GREEN
from compare_ctrgen import Pair, flat

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')
STEPS, MORE = 60, 25


def sha(txt):
    return hashlib.sha256(txt.encode()).hexdigest()[:32]


def record(key, red='fsm', green='enterprise'):
    p = Pair(key, STEPS, red, green)
    mine, ref = p.dumps()
    assert mine == ref, 'scenario #1'
    d1 = sha(ref)
    T = STEPS + MORE
    A = np.zeros((T, 5), np.int16)
    O = np.zeros((T + 2, 578), np.uint8)          # row 0: scenario #2, rows 1..STEPS: steps, row STEPS+1: scenario #3, then the rest
    R = np.zeros(T, np.float32)
    D = np.zeros(T, np.uint8)
    masks, dumps = [], [d1]
    arng = np.random.default_rng(key ^ 0xB10E)
    obs, info = p.reset()
    assert p.check('scenario #2', obs, info)
    O[0] = flat(obs)
    masks.append(np.concatenate([np.array(info[f'blue_agent_{b}']['action_mask'], np.uint8) for b in range(5)]))
    dumps.append(sha(p.dumps()[1]))
    row = 1
    for t in range(T):
        if t == STEPS:
            obs, info = p.reset()
            assert p.check('scenario #3', obs, info)
            O[row] = flat(obs); row += 1
            masks.append(np.concatenate([np.array(info[f'blue_agent_{b}']['action_mask'], np.uint8) for b in range(5)]))
            dumps.append(sha(p.dumps()[1]))
        A[t] = [arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(242)]
        obs, rew, term = p.step(A[t].astype(np.int32))
        assert p.check(f'step {t}', obs, None, rew, term)
        O[row] = flat(obs); row += 1
        R[t] = rew['blue_agent_0']; D[t] = term['blue_agent_0']
    assert D[STEPS - 2] == 1 and D[STEPS - 1] == 1          # done from step count steps - 1 on (ESG.py:870)
    name = f"ctrgen_key{key & 0xFFFFFFFF}_{red}_{green}.npz"
    np.savez_compressed(os.path.join(OUT, name), key=np.uint64(key), steps=np.int32(STEPS), more=np.int32(MORE), actions=A, obs=O, reward=R, done=D,
                        masks=np.stack(masks), dumps=np.array(dumps), red_policy=np.int32(RED[red][1]), green_policy=np.int32(GREEN[green][1]),
                        numpy_version=np.bytes_(np.__version__), proxy_calls=np.bytes_(repr(sorted(p.proxy.calls.items()))))
    print(name, 'sum reward', float(R.sum()), 're-draws', p.proxy.calls.get('gen_redraw'), flush=True)


if __name__ == '__main__':
    record(101); record(102, red='random'); record(103, red='discovery'); record(104, green='sleep'); record(105, red='sleep')
    record((1 << 64) - 59); record(107); record(108, red='random', green='sleep'); record(109); record(110, red='discovery')
