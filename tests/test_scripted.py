"""The reference's own scripted tests as fixtures: tests/golden/scripted_*.json, recorded from the REAL reference by
oracle/refgen/make_scripted_golden.py -- test_blocking_red.py, test_blue_actions.py, test_BlueRewardMachine.py (every subnet x mission
phase), test_Red/{test_Impact, test_DegradeServices, test_DiscoverDeception, test_Withdraw, test_RedSessionCheck}.py, test_Green/{test_GreenLocalWork,
test_GreenAccessService}.py, test_issue22_blocks.py, test_issue26_monitor.py, test_session_issues.py, test_Acceptance/{test_priority, test_challenge_details,
test_deception, test_green_agents}.py, plus a scripted red agent among live FSM agents.  A fixture is a list of cases; a case is (seed, number of resets, script of cc4_edit_state ops and
cc4_step_ex inputs) with, per entry, what the reference did: flat observations, team reward, BlueRewardMachine component, done,
numpy stream position, `success` of every submitting agent, active red agents, malware files per host, and a digest of the
canonical dump of the whole simulator state.

CPU: every case on the oracle.  GPU: every case on the numpy-stream kernel against the fixture, bit for bit; on the two
counter-mode kernels (different streams: the recorded bits do not apply) against the oracle in the same mode, step by step, with
the scenario taken from the numpy stream so that the script's hosts exist."""
import glob
import hashlib
import json
import os
import numpy as np
import pytest
from oracle_binding import OracleVecEnv

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
FILES = sorted(glob.glob(os.path.join(GOLDEN, 'scripted_*.json')))
NAMES = [os.path.basename(p)[len('scripted_'):-len('.json')] for p in FILES]


def _load(name):
    with open(os.path.join(GOLDEN, f'scripted_{name}.json')) as f:
        return json.load(f)


def _sha(txt):
    return hashlib.sha256(txt.encode()).hexdigest()[:32]


def _records(env, entries, kind):
    arr = env.agent_actions(kind)
    for k, t, host, arg, ticks, sid, flags, r0, r1 in entries:
        rec = arr[0, k]
        rec['type'] = t; rec['host'] = host; rec['arg'] = arg; rec['ticks'] = ticks; rec['session'] = sid; rec['flags'] = flags
        rec['rate0'] = r0; rec['rate1'] = r1
    return arr


def _rng_ok(words, want):
    state, has, uint = int(want[0]), want[1], want[2]
    return (int(words[0]) << 64 | int(words[1])) == state and int(words[4]) == has and (not has or int(words[5]) == uint)


def _files(ts):
    out = {}
    for hd in ts['hosts']:
        names = [n for n, bit in (('cmd.sh', 1), ('escalate.sh', 2)) if hd['files'] & bit]
        if len(names) == 2 and not hd['files'] & 4:
            names.reverse()
        if names:
            out[str(hd['h'])] = names
    return out


def _dedup_last(names):
    out = []
    for n in names:
        if n in out:
            out.remove(n)
        out.append(n)
    return out


def replay(env, case, dump_of, tag):
    """Runs one case on `env` (one episode, numpy-stream mode) and checks every recorded expectation.  dump_of(env) -> canonical dump text."""
    env.reset(seeds=np.array([case['seed']], np.uint64))
    for _ in range(1 + case['resets']):
        env.reset(seeds=None)
    n_steps = 0
    for k, e in enumerate(case['script']):
        where = (tag, case.get('note', ''), k)
        if 'edit' in e:
            assert env.edit_state(0, *e['edit']) == e['rc'], where
            assert _sha(dump_of(env)) == e['dump'], where
            assert _rng_ok(env.rng_state()[0], e['rng']), where
            continue
        st, ex = e['step'], e['expect']
        obs, rew, done, info = env.step_ex(np.array([st['blue']], np.int32), None, _records(env, st['red'], 'red'), _records(env, st['green'], 'green'))
        n_steps += 1
        digits = ''.join(str(int(v)) for v in obs[0])
        assert (digits == ex['obs']) if 'obs' in ex else (_sha(digits) == ex['obs_sha']), where       # long trajectories carry a digest
        assert float(rew[0]) == ex['reward'] and int(bool(done[0])) == ex['done'], where
        assert not info['err'].any(), where
        assert _rng_ok(env.rng_state()[0], ex['rng']), where
        assert _sha(dump_of(env)) == ex['dump'], where
        ts = json.loads(env.true_state_json(0))
        assert float(ts['reward'] - ts['action_cost']) == ex['brm'], where
        for agent, want in ex['success'].items():
            i = int(agent.split('_')[-1])
            if agent.startswith('red'):
                assert (ts['red'][i]['obs_success'] or 2) == want, where + (agent,)
            elif agent.startswith('green'):
                sleeping = not any(r[0] == i and r[1] in (0, 1, 3) for r in st['green'])
                assert ((ts['green_fail'][i >> 5] >> (i & 31)) & 1) == int(want == 3 and not sleeping), where + (agent,)
        assert [r for r in range(6) if ts['red'][r]['active']] == ex['active_red'], where
        if ex.get('red_obs'):            # the dict observation the submitting red agents got back (true_state.red_observations), in canonical form
            from cage_challenge_4_amd.true_state import decode, red_observations, red_obs_skeleton
            mine = red_observations(decode(ts))
            for agent, want in ex['red_obs'].items():
                assert red_obs_skeleton(mine[agent]) == want, where + (agent,)
        assert _files(ts) == {h: _dedup_last(v) for h, v in ex['files'].items()}, where
    return n_steps


def test_there_are_enough_fixtures():
    cases = sum(len(_load(n)['cases']) for n in NAMES)
    assert len(NAMES) >= 31 and cases >= 130, (len(NAMES), cases)
    src = ' '.join(_load(n)['source'] for n in NAMES)
    for f in ('test_blocking_red.py', 'test_blue_actions.py', 'test_BlueRewardMachine.py', 'test_Impact.py', 'test_DegradeServices.py',
              'test_DiscoverDeception.py', 'test_Withdraw.py', 'test_RedSessionCheck.py', 'test_GreenLocalWork.py', 'test_GreenAccessService.py',
              'test_issue26_monitor.py', 'test_issue22_blocks.py', 'test_session_issues.py', 'test_priority.py', 'test_challenge_details.py',
              'test_deception.py', 'test_green_agents.py', 'test_mission_phase.py'):
        assert f in src, f


@pytest.mark.parametrize('name', NAMES)
def test_oracle_replays_the_reference_scripted_tests(name):
    fx = _load(name)
    total = 0
    for case in fx['cases']:
        env = OracleVecEnv(1, steps=case['steps'], red_policy=case['red_policy'], green_policy=case['green_policy'])
        total += replay(env, case, lambda e: e.dump(0), name)
        env.close()
    assert total > 0


# ---------------------------------------------------------------------------------------------------------------- GPU
def _device_dump(scratch):
    def fn(env):
        scratch.restore(0, env.snapshot(0))
        return scratch.dump(0)
    return fn


@pytest.mark.gpu
@pytest.mark.parametrize('name', NAMES)
def test_hip_replays_the_reference_scripted_tests(name):
    """The numpy-stream kernel (bit-exact with the reference under the same seed) against the fixture itself, every expectation; the
    canonical dump is taken from the device's packed rows (cc4_get_state + cc4_get_cold) through the oracle's dumper."""
    from cage_challenge_4_amd import CC4VecEnv
    fx = _load(name)
    for case in fx['cases']:
        env = CC4VecEnv(1, steps=case['steps'], red_policy=case['red_policy'], green_policy=case['green_policy'])
        assert env.step_kernel == 'k_step'
        scratch = OracleVecEnv(1, steps=case['steps'], red_policy=case['red_policy'], green_policy=case['green_policy'])
        replay(env, case, _device_dump(scratch), name)
        env.close(); scratch.close()


@pytest.mark.gpu
@pytest.mark.parametrize('lean', ['0', '1'], ids=['4wave', '1wave'])
def test_hip_counter_mode_runs_the_scripted_tests_like_the_oracle(lean, monkeypatch):
    """Both counter-mode kernels: every case of every fixture in ONE batch, scenario from the numpy stream of the case's seed
    (snapshot moved to a counter-mode handle, as for the ctrstep fixtures), the script's edits and steps applied to the device and to
    the oracle in the same mode: observations, rewards, dones, flags after every step, the packed rows at the end."""
    from cage_challenge_4_amd import CC4VecEnv
    monkeypatch.setenv('CC4_PHILOX_LEAN', lean)
    cases = [c for n in NAMES for c in _load(n)['cases']]
    for steps in sorted({c['steps'] for c in cases}):
        group = [c for c in cases if c['steps'] == steps]
        n = len(group)
        envs = []
        for cls in (CC4VecEnv, OracleVecEnv):
            ctr = cls(n, steps=steps, rng_mode=1, **({'strict': False} if cls is CC4VecEnv else {}))
            ctr.reset(seeds=1)
            for i, c in enumerate(group):
                g = cls(1, steps=steps, rng_mode=0, red_policy=c['red_policy'], green_policy=c['green_policy'])
                g.reset(seeds=np.array([c['seed']], np.uint64))
                for _ in range(1 + c['resets']):
                    g.reset(seeds=None)
                ctr.restore(i, g.snapshot(0))
                g.close()
            ctr.set_seed(np.arange(n, dtype=np.uint64) + np.uint64(777))
            envs.append(ctr)
        dev, ora = envs
        assert dev.step_kernel == ('k_step_philox', 'k_step_philox1')[int(lean)]
        T = max(len(c['script']) for c in group)
        finished = np.zeros(n, bool)       # episodes that reached their last step (a script that sets the step count by hand, plus the Sleep
        for k in range(T):                 # steps its edit entries cost here): regenerated, the rest of the script dropped
            blue = np.full((n, 5), -1, np.int32)
            red, green = dev.agent_actions('red'), dev.agent_actions('green')
            stepping = np.zeros(n, bool)
            for i, c in enumerate(group):
                if k >= len(c['script']) or finished[i]:
                    continue
                e = c['script'][k]
                if 'edit' in e:
                    assert dev.edit_state(i, *e['edit']) == ora.edit_state(i, *e['edit'])
                else:
                    stepping[i] = True
                    blue[i] = e['step']['blue']
                    for kind, arr in (('red', red), ('green', green)):
                        for a, t, host, arg, ticks, sid, flags, r0, r1 in e['step'][kind]:
                            rec = arr[i, a]
                            rec['type'] = t; rec['host'] = host; rec['arg'] = arg; rec['ticks'] = ticks; rec['session'] = sid
                            rec['flags'] = flags; rec['rate0'] = r0; rec['rate1'] = r1
            # (episodes whose entry k is an edit, or whose script is over, take a step of Sleeps: the batch moves as one)
            d = dev.step_ex(blue, None, red, green); o = ora.step_ex(blue, None, red, green)
            bad = np.nonzero((d[0] != o[0]).any(axis=1) | (d[1] != o[1]) | (d[2] != o[2]) | (d[3]['err'] != o[3]['err']))[0]
            assert bad.size == 0, (steps, k, bad[:10].tolist())
            if d[2].any():
                over = np.asarray(d[2], bool)
                assert all(k + 3 >= len(c['script']) for c, f in zip(group, over) if f)
                finished |= over
                assert np.array_equal(dev.reset(seeds=None, env_mask=over), ora.reset(seeds=None, env_mask=over))
            if k + 2 >= steps:
                break
        for i in range(n):
            (h1, c1), (h2, c2) = dev.snapshot(i), ora.snapshot(i)
            assert np.array_equal(h1, h2) and np.array_equal(c1, c2), (steps, i)
        dev.close(); ora.close()
