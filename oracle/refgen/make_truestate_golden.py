"""Golden true-state checkpoints from the REAL reference (CybORG.get_true_state), for tests/test_true_state.py.

Replays the blue actions of an existing trajectory fixture (tests/golden/traj_seed123_random_ctor_500.npz) through the
reference and records, at a few steps, a canonical reduction of `env.get_true_state(all hosts, all fields)`:
per hostname the ip / subnet, the processes (PID, kind, root), the services (active, reliability, PID) and the
sessions (agent, id, PID, kind, root).  Data only; runs only where /root/reference exists.

usage: python make_truestate_golden.py     # writes tests/golden/truestate_seed123.json
"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Agents import SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent
from CybORG.Agents.Wrappers import BlueFlatWrapper
from ref_dump import KIND

OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')
ALL = {'Interfaces': 'All', 'Processes': 'All', 'Sessions': 'All', 'Files': 'All', 'User info': 'All', 'System info': 'All', 'Services': 'All'}
CHECKPOINTS = (0, 1, 12, 60, 170, 340, 499)


def kind_of(name):
    if name in KIND:
        return KIND[name]
    return KIND[getattr(name, 'name', name)] if getattr(name, 'name', name) in KIND else KIND[str(name)]


def canon(env):
    st = env.environment_controller.state
    hosts = list(st.hosts.keys())
    ts = env.get_true_state({h: ALL for h in hosts})
    d = ts.data if hasattr(ts, 'data') else dict(ts)
    out = {}
    for h in hosts:
        e = d[h]
        procs = sorted([int(p['PID']), kind_of(p.get('process_name')), int(p.get('username') in ('root', 'SYSTEM'))] for p in e.get('Processes', []))
        sess = sorted([s['agent'], int(s['session_id']), int(s['PID']), str(getattr(s['Type'], 'name', s['Type'])), int(s.get('username') in ('root', 'SYSTEM'))]
                      for s in e.get('Sessions', []))
        svcs = {str(KIND[k]): [int(v.active), int(v._percent_reliable), int(v.process)] for k, v in st.hosts[h].services.items()}
        out[h] = {'ip': str(e['Interface'][0]['ip_address']), 'subnet': str(e['Interface'][0]['Subnet']), 'procs': procs, 'sessions': sess, 'services': svcs}
    blocks = {str(k.value if hasattr(k, 'value') else k): sorted(str(getattr(f, 'value', f)) for f in v) for k, v in st.blocks.items() if v}
    return {'step': int(env.environment_controller.step_count), 'phase': int(st.mission_phase), 'hosts': out, 'blocks': blocks}


def main():
    z = np.load(os.path.join(OUT, 'traj_seed123_random_ctor_500.npz'))
    A = z['actions'].astype(int)
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=500)
    env = CybORG(sg, seed=123)
    w = BlueFlatWrapper(env)
    w.reset()
    cps = {}
    if 0 in CHECKPOINTS:
        cps['0'] = canon(env)
    for t in range(len(A)):
        w.step({f'blue_agent_{b}': int(A[t, b]) for b in range(5)})
        if t + 1 in CHECKPOINTS:
            cps[str(t + 1)] = canon(env)
    doc = {'fixture': 'traj_seed123_random_ctor_500.npz', 'numpy_version': np.__version__, 'checkpoints': cps,
           'note': 'keys of checkpoints = number of steps taken; procs [PID, kind, root]; sessions [agent, id, PID, SessionType name, root]; '
                   'services kind -> [active, reliability, PID]; kind indices as in csrc/cc4_state.h K_*'}
    with open(os.path.join(OUT, 'truestate_seed123.json'), 'w') as f:
        json.dump(doc, f, separators=(',', ':'), sort_keys=True)
    print('wrote truestate_seed123.json', {k: len(v['hosts']) for k, v in cps.items()})


if __name__ == '__main__' and len(sys.argv) == 1:
    main()


def last_actions():
    """tests/golden/lastaction_seed123.json: str() of CybORG.get_last_action for every blue and red agent, 200 steps."""
    z = np.load(os.path.join(OUT, 'traj_seed123_random_ctor_500.npz'))
    A = z['actions'].astype(int)
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=500)
    env = CybORG(sg, seed=123)
    w = BlueFlatWrapper(env)
    w.reset()
    agents = [f'blue_agent_{b}' for b in range(5)] + [f'red_agent_{r}' for r in range(6)]
    rows = []
    for t in range(200):
        w.step({f'blue_agent_{b}': int(A[t, b]) for b in range(5)})
        row = []
        for ag in agents:
            la = env.get_last_action(ag)
            la = la if isinstance(la, list) else [la]
            row.append(' | '.join(str(a) for a in la))
        rows.append(row)
    with open(os.path.join(OUT, 'lastaction_seed123.json'), 'w') as f:
        json.dump({'fixture': 'traj_seed123_random_ctor_500.npz', 'agents': agents, 'steps': rows}, f, separators=(',', ':'))
    print('wrote lastaction_seed123.json', len(rows))


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'lastaction':
    last_actions()
