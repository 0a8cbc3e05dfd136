"""True-state export: cc4_get_true_state + cage_challenge_4_amd.true_state against checkpoints recorded
from the reference's CybORG.get_true_state (oracle/refgen/make_truestate_golden.py -> tests/golden/truestate_seed123.json).
CPU: the shared serializer through the oracle build; GPU: the same through the C ABI of the HIP library."""
import json
import os

import numpy as np
import pytest

import golden_util
from cage_challenge_4_amd import true_state as T

GOLD = os.path.join(golden_util.GOLDEN_DIR, 'truestate_seed123.json')


def canon(ts):
    """Reduce a decoded TrueState to the golden's canonical form."""
    kind = {n: i for i, n in enumerate(T.KIND_NAME)}
    out = {}
    for name, e in ts.hosts.items():
        procs = sorted([p['PID'], kind[p['process_name']] if 'process_name' in p else 9, int(p.get('username') == 'root')] for p in e['Processes'])
        sess = []
        for s in e.get('Sessions', []):
            sess.append([s['agent'], s['session_id'], s['PID'], s['Type'], int(s['username'] == 'root')])
        svcs = {str(kind[k]): [int(v['active']), v['reliability'], v['PID']] for k, v in e['Services'].items()}
        out[name] = {'ip': str(e['Interface'][0]['ip_address']), 'subnet': str(e['Interface'][0]['Subnet']), 'procs': procs,
                     'sessions': sorted(sess), 'services': svcs}
    return out


def check(make_env):
    doc = json.load(open(GOLD))
    fix = golden_util.load(os.path.join(golden_util.GOLDEN_DIR, doc['fixture']))
    env = make_env(fix)
    cps = {int(k): v for k, v in doc['checkpoints'].items()}
    for t in range(len(fix['actions']) + 1):
        if t in cps:
            ts = T.decode(env.true_state_json(0))
            want = cps[t]
            assert ts.step == want['step'] and ts.phase == want['phase'], t
            got = canon(ts)
            assert set(got) == set(want['hosts']), t
            for h, w in want['hosts'].items():
                g = got[h]
                assert g['ip'] == w['ip'] and g['subnet'] == w['subnet'], (t, h)
                assert g['procs'] == [list(p) for p in w['procs']], (t, h, g['procs'], w['procs'])
                assert g['services'] == {k: list(v) for k, v in w['services'].items()}, (t, h)
                # session type: the engine tracks "RedAbstractSession or not" for red sessions; blue/green types are fixed
                gs = [[a, i, p, ty == 'RED_ABSTRACT_SESSION', r] if a.startswith('red') else [a, i, p] for a, i, p, ty, r in g['sessions']]
                ws = [[a, i, p, ty == 'RED_ABSTRACT_SESSION', r] if a.startswith('red') else [a, i, p] for a, i, p, ty, r in w['sessions']]
                assert gs == ws, (t, h, gs, ws)
            want_access = {}
            for h, w in want['hosts'].items():
                lv = [r for a, i, p, ty, r in w['sessions'] if a.startswith('red')]
                if lv:
                    want_access[h] = 'root' if any(lv) else 'user'
            assert ts.red_access() == want_access, t
            blocks = {k: sorted(v) for k, v in ts.blocks.items()}
            assert blocks == want['blocks'], (t, blocks, want['blocks'])
        if t < len(fix['actions']):
            env.step(fix['actions'][t][None, :])


def test_true_state_matches_reference_checkpoints_oracle_build():
    from oracle_binding import OracleVecEnv

    def make(fix):
        e = OracleVecEnv(1, steps=fix['steps'])
        e.reset(seeds=fix['seed'])
        e.reset(seeds=None)        # CybORG(seed=s); wrapper.reset()
        return e
    check(make)


@pytest.mark.gpu
def test_true_state_matches_reference_checkpoints_hip():
    from cage_challenge_4_amd import CC4VecEnv

    def make(fix):
        e = CC4VecEnv(1, steps=fix['steps'])
        e.reset(seeds=fix['seed'])
        e.reset(seeds=None)
        return e
    check(make)


def test_table_wrapper_over_decoded_state():
    from oracle_binding import OracleVecEnv
    e = OracleVecEnv(1, steps=50)
    e.reset(seeds=7)

    class Env:     # anything with get_true_state()
        def get_true_state(self, info=None):
            return T.decode(e.true_state_json(0)).as_dict(info)
    w = T.TrueStateTableWrapper(Env())
    t = w.get_host_overview_table()
    assert len(t.rows) == len(w.hostnames) and 'Hostname' in str(t)
    tabs = w.get_host_processes_tables()
    assert set(tabs) == set(T.SUBNETS)
    assert sum(len(x.rows) for x in tabs.values()) == sum(len(v['Processes']) for k, v in w.get_raw_full_true_state().items() if k != 'success')


def test_cyborg_get_true_state_surface():
    """CybORG.get_true_state / TrueStateTableWrapper over the drop-in classes (oracle backend injected as the vec factory)."""
    from oracle_binding import OracleVecEnv
    from cage_challenge_4_amd import CybORG, EnterpriseScenarioGenerator, TrueStateTableWrapper
    env = CybORG(EnterpriseScenarioGenerator(steps=30), seed=11, vec_factory=OracleVecEnv)
    ts = env.get_true_state()
    assert ts.pop('success') is True
    assert set(ts) == set(env.get_ip_map())
    assert all(str(v['Interface'][0]['ip_address']) == env.get_ip_map()[k] for k, v in ts.items())
    assert 'contractor_network_subnet_user_host_0' in ts or len(ts) > 20
    sub = env.get_true_state({next(iter(ts)): 'All'})
    assert len(sub) == 2
    assert 'Hostname' in str(TrueStateTableWrapper(env).get_host_overview_table())


def _check_last_actions(make_env):
    doc = json.load(open(os.path.join(golden_util.GOLDEN_DIR, 'lastaction_seed123.json')))
    fix = golden_util.load(os.path.join(golden_util.GOLDEN_DIR, doc['fixture']))
    env = make_env(fix)
    for t, want in enumerate(doc['steps']):
        env.step(fix['actions'][t][None, :])
        ts = T.decode(env.true_state_json(0))
        got = [str(ts.last_action[a]) for a in doc['agents']]
        assert got == want, (t, [(a, g, w) for a, g, w in zip(doc['agents'], got, want) if g != w])


def test_last_action_matches_reference_oracle_build():
    """CybORG.get_last_action (env.py:300-314): str() of the resolved action of all 11 stateful agents, 200 steps."""
    from oracle_binding import OracleVecEnv

    def make(fix):
        e = OracleVecEnv(1, steps=fix['steps'])
        e.reset(seeds=fix['seed'])
        e.reset(seeds=None)
        return e
    _check_last_actions(make)


@pytest.mark.gpu
def test_last_action_matches_reference_hip():
    from cage_challenge_4_amd import CC4VecEnv

    def make(fix):
        e = CC4VecEnv(1, steps=fix['steps'])
        e.reset(seeds=fix['seed'])
        e.reset(seeds=None)
        return e
    _check_last_actions(make)
