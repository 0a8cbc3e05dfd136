"""Golden blue dict observations from the REAL reference, for tests/test_blue_obs.py.

Runs the reference with SleepAgent blue (every observation is the end-of-turn Monitor's, Monitor.py:35-74) for the seed of
tests/golden/traj_seed123_sleep_ctor_500.npz and records, per step and blue agent, a canonical (JSON) form of
CybORG.get_observation(agent): success, action, and per hostname Interface / Processes / System info.  Data only; runs only
where /root/reference exists.

usage: python make_blueobs_golden.py [steps]   # writes tests/golden/blueobs_seed123.json
"""
import hashlib, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Agents import SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent
from CybORG.Agents.Wrappers import BlueFlatWrapper

OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')
DIGEST_UNTIL = 450      # beyond the fully recorded steps, a 64-bit digest of each canonical observation up to this step


def digest(d):
    return hashlib.sha256(json.dumps(d, sort_keys=True, separators=(',', ':')).encode()).hexdigest()[:16]


def canon(v):
    if isinstance(v, dict):
        return {str(k): canon(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [canon(x) for x in v]
    if isinstance(v, np.ndarray):
        return [canon(x) for x in v.tolist()]
    if isinstance(v, (bool, int, str)) or v is None:
        return v
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (float, np.floating)):
        return float(v)
    return getattr(v, 'name', None) or str(v)


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=500)
    env = CybORG(sg, seed=123)
    w = BlueFlatWrapper(env)
    w.reset()
    rows = []
    for t in range(steps):
        w.step({})
        row = {}
        for b in range(5):
            o = env.get_observation(f'blue_agent_{b}')
            d = canon(o)
            d.pop('message', None)
            row[f'blue_agent_{b}'] = d
        rows.append(row)
    digs = []
    for t in range(steps, DIGEST_UNTIL):
        w.step({})
        row = []
        for b in range(5):
            d = canon(env.get_observation(f'blue_agent_{b}'))
            d.pop('message', None)
            row.append(digest(d))
        digs.append(row)
    with open(os.path.join(OUT, 'blueobs_seed123.json'), 'w') as f:
        json.dump({'fixture': 'traj_seed123_sleep_ctor_500.npz', 'steps': rows, 'digests': digs, 'numpy_version': np.__version__}, f, separators=(',', ':'), sort_keys=True)
    import collections
    c = collections.Counter()
    for r in rows:
        for a, d in r.items():
            for h, e in d.items():
                if isinstance(e, dict):
                    for p in e.get('Processes', []):
                        c[tuple(sorted(p.keys())) + tuple(sorted(k for cn in p.get('Connections', []) for k in cn))] += 1
    for k, v in c.most_common():
        print(v, k)
    print('wrote blueobs_seed123.json', len(rows), os.path.getsize(os.path.join(OUT, 'blueobs_seed123.json')))


def random_blue(steps=120):
    """tests/golden/blueobs_seed123_random.json: the same with the random blue actions of traj_seed123_random_ctor_500.npz.
    Per step and agent: success, action string, and the host entries (the Monitor part of the observation)."""
    z = np.load(os.path.join(OUT, 'traj_seed123_random_ctor_500.npz'))
    A = z['actions'].astype(int)
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=500)
    env = CybORG(sg, seed=123)
    w = BlueFlatWrapper(env)
    w.reset()
    rows = []
    for t in range(steps):
        w.step({f'blue_agent_{b}': int(A[t, b]) for b in range(5)})
        row = {}
        for b in range(5):
            o = env.get_observation(f'blue_agent_{b}')
            act = o.get('action')
            d = canon({k: v for k, v in o.items() if k not in ('message', 'action')})
            d['action'] = None if act is None else str(act)
            row[f'blue_agent_{b}'] = d
        rows.append(row)
    digs = []
    for t in range(steps, DIGEST_UNTIL):
        w.step({f'blue_agent_{b}': int(A[t, b]) for b in range(5)})
        row = []
        for b in range(5):
            o = env.get_observation(f'blue_agent_{b}')
            act = o.get('action')
            d = canon({k: v for k, v in o.items() if k not in ('message', 'action')})
            d['action'] = None if act is None else str(act)
            row.append(digest(d))
        digs.append(row)
    with open(os.path.join(OUT, 'blueobs_seed123_random.json'), 'w') as f:
        json.dump({'fixture': 'traj_seed123_random_ctor_500.npz', 'steps': rows, 'digests': digs, 'numpy_version': np.__version__}, f, separators=(',', ':'), sort_keys=True)
    print('wrote blueobs_seed123_random.json', len(rows), os.path.getsize(os.path.join(OUT, 'blueobs_seed123_random.json')))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'random':
        random_blue()
    else:
        main()
