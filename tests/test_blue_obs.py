"""Blue dict observations decoded from the per-step event log against the reference's own
CybORG.get_observation for 300 steps (oracle/refgen/make_blueobs_golden.py -> tests/golden/blueobs_seed123.json)."""
import hashlib
import json
import os

import numpy as np
import pytest

import golden_util
from cage_challenge_4_amd import true_state as T

GOLD = os.path.join(golden_util.GOLDEN_DIR, 'blueobs_seed123.json')


def _canon(v):
    if isinstance(v, dict):
        return {str(k): _canon(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_canon(x) for x in v]
    if isinstance(v, (bool, int, float, str)) or v is None:
        return v
    return str(v)


def _digest(d):
    return hashlib.sha256(json.dumps(d, sort_keys=True, separators=(',', ':')).encode()).hexdigest()[:16]


def _check(make_env):
    doc = json.load(open(GOLD))
    fix = golden_util.load(os.path.join(golden_util.GOLDEN_DIR, doc['fixture']))
    env = make_env(fix)
    env.enable_event_log(True)
    for t, want in enumerate(doc['steps']):
        env.step(np.full((1, 5), -1, np.int32))
        got = T.blue_observations(T.decode(env.true_state_json(0)))
        for agent, w in want.items():
            g = _canon(got[agent])
            assert g == w, (t, agent, {k: (g.get(k), w.get(k)) for k in set(g) | set(w) if g.get(k) != w.get(k)})
    for i, row in enumerate(doc['digests']):        # steps 200..449: a digest of each canonical observation
        env.step(np.full((1, 5), -1, np.int32))
        got = T.blue_observations(T.decode(env.true_state_json(0)))
        for b in range(5):
            assert _digest(_canon(got[f'blue_agent_{b}'])) == row[b], (len(doc['steps']) + i, b)


def test_blue_dict_observations_match_reference_oracle_build():
    from oracle_binding import OracleVecEnv

    def make(fix):
        e = OracleVecEnv(1, steps=fix['steps'])
        e.reset(seeds=fix['seed'])
        e.reset(seeds=None)
        return e
    _check(make)


@pytest.mark.gpu
def test_blue_dict_observations_match_reference_hip():
    from cage_challenge_4_amd import CC4VecEnv

    def make(fix):
        e = CC4VecEnv(1, steps=fix['steps'])
        e.reset(seeds=fix['seed'])
        e.reset(seeds=None)
        return e
    _check(make)


def _check_random(cls):
    """The same under random blue actions (fixture traj_seed123_random_ctor_500): 'action' string, 'success', and the
    host entries, incl. the process entry of a resolved DeployDecoy and the file list of a resolved Analyse."""
    doc = json.load(open(os.path.join(golden_util.GOLDEN_DIR, 'blueobs_seed123_random.json')))
    fix = golden_util.load(os.path.join(golden_util.GOLDEN_DIR, doc['fixture']))
    e = cls(1, steps=fix['steps'])
    e.reset(seeds=fix['seed'])
    e.reset(seeds=None)
    e.enable_event_log(True)
    checked = 0
    for t, want in enumerate(doc['steps']):
        e.step(fix['actions'][t][None, :])
        got = T.blue_observations(T.decode(e.true_state_json(0)))
        for agent, w in want.items():
            g = _canon(got[agent])
            assert g.get('action') == w['action'], (t, agent)
            assert g['success'] == w['success'], (t, agent)
            hosts_g = {k: v for k, v in g.items() if k not in ('success', 'action')}
            hosts_w = {k: v for k, v in w.items() if k not in ('success', 'action')}
            assert hosts_g == hosts_w, (t, agent)
            checked += 1
    assert checked > 550
    t0 = len(doc['steps'])
    for i, row in enumerate(doc['digests']):        # steps 120..449 by digest (covers late-episode Restore/Analyse/decoys)
        e.step(fix['actions'][t0 + i][None, :])
        got = T.blue_observations(T.decode(e.true_state_json(0)))
        for b in range(5):
            g = _canon(got[f'blue_agent_{b}'])
            g.setdefault('action', None)
            assert _digest(g) == row[b], (t0 + i, b)


def test_blue_dict_observations_with_random_blue_actions_oracle_build():
    from oracle_binding import OracleVecEnv
    _check_random(OracleVecEnv)


@pytest.mark.gpu
def test_blue_dict_observations_with_random_blue_actions_hip():
    """the device's event log under random blue actions against the same recording of the reference."""
    from cage_challenge_4_amd import CC4VecEnv
    _check_random(CC4VecEnv)


def test_cyborg_get_observation_surface():
    from oracle_binding import OracleVecEnv
    from cage_challenge_4_amd import CybORG, EnterpriseScenarioGenerator, EnterpriseGreenAgent, FiniteStateRedAgent, SleepAgent, BlueFlatWrapper
    doc = json.load(open(GOLD))
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent, red_agent_class=FiniteStateRedAgent, steps=500)
    env = CybORG(sg, seed=123, vec_factory=OracleVecEnv)
    w = BlueFlatWrapper(env)
    w.reset()
    for t in range(12):
        w.step({})
        for b in range(5):
            assert _canon(env.get_observation(f'blue_agent_{b}')) == doc['steps'][t][f'blue_agent_{b}'], (t, b)
