"""tests/golden/facade_seed123.json: the raw CybORG surface around the step, recorded from the REAL reference
(CybORG/env.py:95-161, 316-372, 405-415): `parallel_step` fed with Action OBJECTS, its per-agent reward components and
done flags, `get_rewards`, `active_agents`, `get_last_action` of the red agents (sensitive to every RNG draw), and a
`set_seed` in the middle of the episode.

usage: python make_facade_golden.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Agents import SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent
from CybORG.Agents.Wrappers import BlueFlatWrapper

OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')
SEED, STEPS, T, RESEED_AT, RESEED = 123, 120, 90, 40, 77


def describe(a):
    """[class name, {parameter: value}] of a reference Action object (what the mirror's action classes take)."""
    d = {}
    for k in ('hostname', 'from_subnet', 'to_subnet'):
        if hasattr(a, k):
            d[k] = str(getattr(a, k))
    return [type(a).__name__, d]


def main():
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent,
                                     red_agent_class=FiniteStateRedAgent, steps=STEPS)
    env = CybORG(sg, seed=SEED)
    w = BlueFlatWrapper(env)          # only to borrow the reference's own Action objects (its fixed action lists)
    w.reset()
    arng = np.random.default_rng(SEED ^ 0xFACADE)
    rows = []
    for t in range(T):
        if t == RESEED_AT:
            env.set_seed(RESEED)
        acts, desc = {}, []
        for b in range(5):
            ag = f'blue_agent_{b}'
            lst = w._action_space[ag]['actions']
            labels = w._action_space[ag]['labels']
            valid = [i for i, l in enumerate(labels) if not l.startswith('[')]
            i = int(valid[arng.integers(len(valid))])
            if arng.random() < 0.5:
                acts[ag] = lst[i]
            desc.append([i] + describe(lst[i]) + [ag in acts])
        obs, rew, done, info = env.parallel_step(dict(acts), messages=None)
        reds = {f'red_agent_{r}': ' | '.join(str(a) for a in env.get_last_action(f'red_agent_{r}')) for r in range(6)}
        blues = {f'blue_agent_{b}': ' | '.join(str(a) for a in env.get_last_action(f'blue_agent_{b}')) for b in range(5)}
        compact = lambda names: {'blue': sum(n.startswith('blue') for n in names), 'green': sum(n.startswith('green') for n in names),   # noqa: E731
                                 'red': sorted(n for n in names if n.startswith('red'))}
        rows.append({'actions': desc,
                     'returned_agents': compact(obs.keys()),
                     'rewards_blue': {a: rew[a] for a in sorted(rew) if a.startswith('blue')},
                     'rewards_other': sorted({json.dumps(rew[a], sort_keys=True) for a in rew if not a.startswith('blue')}),
                     'dones': sorted(set(done.values())),
                     'get_rewards': env.get_rewards(),
                     'active_agents': compact(env.active_agents),
                     'last_red': reds, 'last_blue': blues})
    with open(os.path.join(OUT, 'facade_seed123.json'), 'w') as f:
        json.dump({'seed': SEED, 'steps': STEPS, 'reseed_at': RESEED_AT, 'reseed': RESEED, 'rows': rows}, f, separators=(',', ':'))
    print('wrote facade_seed123.json', len(rows), 'steps; reward sum', sum(sum(r['rewards_blue']['blue_agent_0'].values()) for r in rows))


if __name__ == '__main__':
    main()
