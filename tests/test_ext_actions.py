"""Externally submitted red / green actions (cc4_step_ex; SimulationController.py:236-240: the step takes `actions[agent]` for
ANY agent) and `action.duration` on blue indices.  The pin to the reference is oracle/refgen/compare_ext.py (live differential
runs) and the scripted fixtures (tests/test_scripted.py); here: host-side properties on the oracle (CPU) and HIP == oracle for
seeded random submissions on all three step kernels (GPU)."""
import json
import numpy as np
import pytest
from oracle_binding import OracleVecEnv, random_actions
import ext_util as X


def _docs(env):
    return [env.true_state_json(i) for i in range(env.num_envs)]


@pytest.mark.parametrize('rng_mode', [0, 1], ids=['pcg64', 'philox'])
def test_a_step_that_submits_nothing_is_the_plain_step(rng_mode):
    """Every record CC4_ACT_NONE == cc4_step: the policies act and draw exactly as without the records."""
    n, T = 6, 120
    a = OracleVecEnv(n, steps=100, rng_mode=rng_mode, autoreset=True); b = OracleVecEnv(n, steps=100, rng_mode=rng_mode, autoreset=True)
    a.reset(seeds=50); b.reset(seeds=50)
    red, green = b.agent_actions('red'), b.agent_actions('green')
    for t in range(T):
        acts = random_actions(50, t, n)
        if b._done.all():                              # (step_ex has no autoreset of its own in the oracle binding)
            a.step(acts); b.step(acts)
            continue
        oa = a.step(acts); ob = b.step_ex(acts, None, red, green)
        assert np.array_equal(oa[0], ob[0]) and np.array_equal(oa[1], ob[1]) and np.array_equal(oa[2], ob[2]), t
    assert np.array_equal(a.rng_state(), b.rng_state())
    for i in range(n):
        assert np.array_equal(a.get_state(i), b.get_state(i))


def test_a_submitted_red_action_bypasses_the_policy_and_its_draws():
    """red_agent_0 (FiniteStateRedAgent) gets a submitted Sleep every step: its FSM never advances (no step count, no host-state
    change beyond the initial observation), no session ever spreads, and the shared numpy stream moves less than in the plain run."""
    a = OracleVecEnv(1, steps=100, green_policy=1); b = OracleVecEnv(1, steps=100, green_policy=1)   # (no green agents: no phishing)
    a.reset(seeds=7); b.reset(seeds=7)
    red = b.agent_actions('red')
    red['type'][0, 0] = 9                                              # Sleep
    for t in range(60):
        a.step(None); b.step_ex(None, None, red, None)
    da, db = json.loads(a.true_state_json(0)), json.loads(b.true_state_json(0))
    assert sum(len(r['sessions']) for r in db['red']) == 1 and sum(len(r['sessions']) for r in da['red']) > 1
    assert b.rng_state()[0][6] < a.rng_state()[0][6]                   # fewer generator advances
    assert not b._err.any()


def test_invalid_and_unknown_parameters_resolve_as_invalid_action():
    """An address the agent's ActionSpace does not hold, an unknown session, a class outside its action space: InvalidAction --
    one tick, success FALSE (SimulationController.py:1085-1110; InvalidAction.execute, Action.py:59-60)."""
    o = OracleVecEnv(1, steps=50); o.reset(seeds=3)
    d = json.loads(o.true_state_json(0))
    start = d['red'][0]['start']
    unknown = next(h['h'] for h in d['hosts'] if h['h'] // 17 == 0 and h['h'] % 17 >= 11)       # a restricted-zone-A server: not discovered yet
    for rec in ({'type': 1, 'host': unknown, 'session': 0}, {'type': 1, 'host': start, 'session': 5}, {'type': 10}):
        red = o.agent_actions('red')
        for k, v in rec.items():
            red[k][0, 0] = v
        o.step_ex(None, None, red, None)
        d = json.loads(o.true_state_json(0))
        assert d['red'][0]['obs_success'] == 3 and d['last_red'][0][0] == 10, rec
        assert d['red'][0]['busy'] == 0
    # the same scan with a known address and the agent's own session resolves (AggressiveServiceDiscovery: one tick)
    red = o.agent_actions('red')
    red['type'][0, 0] = 1; red['host'][0, 0] = start; red['session'][0, 0] = 0
    o.step_ex(None, None, red, None)
    d = json.loads(o.true_state_json(0))
    assert d['red'][0]['obs_success'] == 1 and d['last_red'][0][0] == 1


def test_duration_override_shortens_and_lengthens_actions():
    o = OracleVecEnv(1, steps=50); o.reset(seeds=3)
    d = json.loads(o.true_state_json(0))
    start = d['red'][0]['start']
    red = o.agent_actions('red')
    red['type'][0, 0] = 4; red['host'][0, 0] = start; red['session'][0, 0] = 0; red['ticks'][0, 0] = 1     # ExploitRemoteService: 4 ticks by default
    o.step_ex(None, None, red, None)
    d = json.loads(o.true_state_json(0))
    assert d['red'][0]['busy'] == 0 and d['last_red'][0][0] == 4 and d['red'][0]['obs_success'] in (1, 3)
    red['type'][0, 0] = 0; red['arg'][0, 0] = 4; red['ticks'][0, 0] = 3                                     # DiscoverRemoteSystems over 3 ticks
    o.step_ex(None, None, red, None)
    none = o.agent_actions('red')
    for k in range(2):
        d = json.loads(o.true_state_json(0))
        assert d['red'][0]['busy'] == 1 and d['red'][0]['obs_success'] == 4                                 # IN_PROGRESS
        o.step_ex(None, None, none, None)
    d = json.loads(o.true_state_json(0))
    assert d['red'][0]['busy'] == 0 and d['red'][0]['obs_success'] == 1 and d['last_red'][0][0] == 0
    # blue: Restore (5 ticks) with duration 1 resolves in the step it is submitted in; the -1 is charged as always
    mask = o.mask()[0]
    acts = np.full((1, 5), -1, np.int32)
    nh = 16
    idx = next(i for i in range(2 * nh + 1, 3 * nh + 1) if mask[i])                                          # a valid Restore slot of blue_agent_0
    acts[0, 0] = idx | (1 << 20)
    o.step_ex(acts, None, none, None)
    d = json.loads(o.true_state_json(0))
    assert d['last_blue'][0][0] == 4 and d['blue'][0]['busy'] == 0 and d['action_cost'] == -1.0


def test_submitted_green_actions_rates_and_validity():
    """fp_detection_rate / phishing_error_rate 0 and 1 (the reference's tests use exactly these); a SleepAgent green agent's action
    space holds Sleep only; GreenAccessService with a subnet outside the agent's current list is invalid."""
    o = OracleVecEnv(1, steps=50, green_policy=1); o.reset(seeds=11)            # green_agent_class=SleepAgent
    d = json.loads(o.true_state_json(0))
    g = 0
    gh = d['green_hosts'][g]
    green = o.agent_actions('green')
    green['type'][0, g] = 1; green['host'][0, g] = gh; green['flags'][0, g] = 1 | 2; green['rate0'][0, g] = 1.0; green['rate1'][0, g] = 0.0
    o.step_ex(None, None, None, green)
    d = json.loads(o.true_state_json(0))
    assert d['green_fail'][0] & 1                                               # not in the action space: InvalidAction -> success False
    assert not any(h['ev'] for h in d['hosts'])
    green['flags'][0, g] |= 4                                                   # skip_valid_action_check=True
    o.step_ex(None, None, None, green)
    d = json.loads(o.true_state_json(0))
    assert not (d['green_fail'][0] & 1)
    ev = {h['h']: h['ev'] for h in d['hosts']}
    assert ev[gh] & 8                                                           # the false-positive process_creation event, rolled over by the end-of-turn Monitor
    # phishing_error_rate 1: a red session appears on the green host
    green['rate0'][0, g] = 0.0; green['rate1'][0, g] = 1.0
    o.step_ex(None, None, None, green)
    d = json.loads(o.true_state_json(0))
    assert any(s[1] == gh for r in d['red'] for s in r['sessions'])
    assert not o._err.any()
    o2 = OracleVecEnv(1, steps=50); o2.reset(seeds=11)
    d = json.loads(o2.true_state_json(0))
    gh = d['green_hosts'][0]
    green = o2.agent_actions('green')
    green['type'][0, 0] = 0; green['host'][0, 0] = gh; green['session'][0, 0] = 0x1FF          # every subnet: more than the agent's own list
    o2.step_ex(None, None, None, green)
    assert json.loads(o2.true_state_json(0))['green_fail'][0] & 1
    green['session'][0, 0] = 1 << (gh // 17)                                                    # its own subnet only: always in the list
    o2.step_ex(None, None, None, green)
    assert not (json.loads(o2.true_state_json(0))['green_fail'][0] & 1)


def test_state_edits():
    """cc4_edit_state ops on the oracle build of the same source (state_edit, csrc/cc4_engine.h)."""
    o = OracleVecEnv(1, steps=50, red_policy=1, green_policy=1); o.reset(seeds=100)
    d = json.loads(o.true_state_json(0))
    h = next(x['h'] for x in d['hosts'] if x['h'] // 17 == 4 and x['h'] % 17 == 11)           # contractor_network_subnet_server_host_0
    before = next(x for x in d['hosts'] if x['h'] == h)
    if not any(s[0] == 1 for s in before['svcs']):
        pid = o.edit_state(0, 1, h, 1, 0)                                                        # add an OT service
        after = next(x for x in json.loads(o.true_state_json(0))['hosts'] if x['h'] == h)
        assert [1, 1, 100, pid] in after["svcs"] and [pid, 13, 0] in after["procs"] and pid > max(p[0] for p in before['procs'])
    o.edit_state(0, 2, h, 0)
    after = next(x for x in json.loads(o.true_state_json(0))['hosts'] if x['h'] == h)
    assert all(s[2] == 0 for s in after['svcs'])
    o.edit_state(0, 0, 2)
    assert json.loads(o.true_state_json(0))['phase'] == 2
    o.edit_state(0, 3, h)
    after = next(x for x in json.loads(o.true_state_json(0))['hosts'] if x['h'] == h)
    assert after['procs'] == [] and after['svcs'] == []
    assert o.edit_state(0, 4, h, 5) == 1 and o.edit_state(0, 4, h, 5) == 0                     # the second apache decoy finds port 80 taken
    after = next(x for x in json.loads(o.true_state_json(0))['hosts'] if x['h'] == h)
    assert len(after['procs']) == 1 and after['svcs'][0][0] == 5
    with pytest.raises(RuntimeError):
        o.edit_state(0, 99)


# ---------------------------------------------------------------------------------------------------------------- GPU
def _dev(n, **kw):
    from cage_challenge_4_amd import CC4VecEnv
    return CC4VecEnv(n, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize('kernel,rng_mode,lean', [('k_step', 0, None), ('k_step_philox', 1, '0'), ('k_step_philox1', 1, '1')])
@pytest.mark.parametrize('policies', [(0, 0), (3, 0), (1, 1)], ids=['fsm', 'randomselect', 'sleep'])
def test_hip_submitted_actions_match_oracle(kernel, rng_mode, lean, policies, monkeypatch):
    """cc4_step_ex on every step kernel (their full builds) against the oracle: seeded random red / green submissions built from
    the episodes' own state -- valid, invalid, with duration / rate overrides, skip_valid_action_check -- plus blue indices with
    durations; observations, rewards, dones, flags every step, the agents' results (red obs_success and observation keys,
    green_fail) through the true-state document every tenth, the packed state and the cold rows at the end."""
    if lean is not None:
        monkeypatch.setenv('CC4_PHILOX_LEAN', lean)
    rp, gp = policies
    n, T, steps = 48, 130, 100
    dev = _dev(n, steps=steps, rng_mode=rng_mode, red_policy=rp, green_policy=gp, strict=False)   # (an FSM agent fed foreign observations may flag: compared, not raised)
    assert dev.step_kernel == kernel
    ora = OracleVecEnv(n, steps=steps, rng_mode=rng_mode, red_policy=rp, green_policy=gp)
    assert np.array_equal(dev.reset(seeds=606), ora.reset(seeds=606))
    rng = np.random.default_rng(17)
    red, green = dev.agent_actions('red'), dev.agent_actions('green')
    for t in range(T):
        if t == steps - 1:                                            # the episodes are over: regenerate all of them, carry on
            assert np.array_equal(dev.reset(seeds=None), ora.reset(seeds=None))
        X.random_ext(_docs(ora), rng, 0.4, 0.08, red, green)
        acts = X.with_durations(random_actions(606, t, n), rng)
        if t % 7 == 3:                                                # a plain step in between: queued actions keep their rates
            d = dev.step(acts); o = ora.step_ex(acts, None, None, None)
        else:
            d = dev.step_ex(acts, None, red, green); o = ora.step_ex(acts, None, red, green)
        bad = np.nonzero((d[0] != o[0]).any(axis=1) | (d[1] != o[1]) | (d[2] != o[2]) | (d[3]['err'] != o[3]['err']))[0]
        assert bad.size == 0, (t, bad[:10].tolist())
        if t % 10 == 9:
            for i in (0, n // 2, n - 1):
                a_, b_ = json.loads(dev.true_state_json(i)), json.loads(ora.true_state_json(i))
                assert a_['red'] == b_['red'] and a_['green_fail'] == b_['green_fail'] and a_['last_red'] == b_['last_red'], (t, i)
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(n):
        (h1, c1), (h2, c2) = dev.snapshot(i), ora.snapshot(i)
        assert np.array_equal(h1, h2), f'packed state differs env {i} at byte offsets {np.nonzero(h1 != h2)[0][:20].tolist()}'
        assert np.array_equal(c1, c2), f'cold row differs env {i} at byte offsets {np.nonzero(c1 != c2)[0][:20].tolist()}'
    dev.close()


@pytest.mark.gpu
def test_hip_state_edits_match_oracle():
    dev = _dev(2, steps=50, red_policy=1, green_policy=1); ora = OracleVecEnv(2, steps=50, red_policy=1, green_policy=1)
    dev.reset(seeds=100); ora.reset(seeds=100)
    d = json.loads(ora.true_state_json(1))
    h = next(x['h'] for x in d['hosts'] if x['h'] // 17 == 4 and x['h'] % 17 == 11)
    for op, a0, a1, a2 in ((0, 1, 0, 0), (2, h, 40, 0), (4, h, 6, 0), (3, h + 1, 0, 0), (1, h + 1, 1, 1), (4, h + 1, 8, 0)):
        assert dev.edit_state(1, op, a0, a1, a2) == ora.edit_state(1, op, a0, a1, a2)
    for i in range(2):
        assert np.array_equal(dev.get_state(i), ora.get_state(i))
    for t in range(10):
        a = random_actions(9, t, 2)
        d_, o_ = dev.step(a), ora.step(a)
        assert np.array_equal(d_[0], o_[0]) and np.array_equal(d_[1], o_[1])
    dev.close()
