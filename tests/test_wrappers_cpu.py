"""CPU: the host-side mirror of the reference's wrapper/evaluation surface, with the CPU oracle injected as the backend
(`vec_factory`), against golden data recorded from the real reference.  The same tests run on the HIP backend in
tests/test_hip_parity.py."""
import json
import os
import numpy as np
import golden_util as G
from oracle_binding import OracleVecEnv
from cage_challenge_4_amd import (CybORG, EnterpriseScenarioGenerator, SleepAgent, EnterpriseGreenAgent,
                                  FiniteStateRedAgent, BlueFlatWrapper, BlueEnterpriseWrapper, EnterpriseMAE)
from cage_challenge_4_amd.evaluation import run_evaluation


def make_sg(steps=500):
    return EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent,
                                       red_agent_class=FiniteStateRedAgent, steps=steps)


def test_drop_in_wrapper_replays_reference_episode(oracle_lib):
    fix = G.load([p for p in G.list_fixtures() if 'seed123_random_ctor_500' in p][0])
    env = BlueFlatWrapper(CybORG(make_sg(), seed=123, vec_factory=OracleVecEnv))
    obs, info = env.reset()
    assert np.array_equal(np.concatenate([obs[a] for a in env.agents]), fix['obs'][0])
    assert np.array_equal(np.concatenate([info[a]['action_mask'] for a in env.agents]), fix['mask'])
    for t in range(500):
        acts = {f'blue_agent_{b}': int(fix['actions'][t, b]) for b in range(5)}
        obs, rew, term, trunc, info = env.step(acts)
        assert np.array_equal(np.concatenate([obs[a] for a in env.possible_agents]), fix['obs'][t + 1]), t
        assert set(rew.values()) == {float(fix['reward'][t])}
        assert term['blue_agent_3'] == bool(fix['done'][t])
    assert env.agents == []          # every agent is done after the last step (BlueFixedActionWrapper.py:164-166)


def test_labels_masks_spaces_and_maps(oracle_lib):
    cy = CybORG(make_sg(), seed=5, vec_factory=OracleVecEnv)
    env = BlueFlatWrapper(cy)
    env.reset()
    for b, a in enumerate(env.possible_agents):
        n = 242 if b == 4 else 82
        labels, mask = env.action_labels(a), env.action_mask(a)
        assert len(labels) == len(mask) == n == env.action_space(a).n
        for lab, m in zip(labels, mask):
            assert lab.startswith('[Invalid]') == (not m)
        assert sum(l == 'Monitor' for l in labels) == 1 and sum(l == 'Sleep' for l in labels) == 1
    l0 = env.action_labels('blue_agent_0')
    assert l0[0] == 'Analyse restricted_zone_a_subnet_server_host_0'
    assert l0[50].startswith('AllowTrafficZone restricted_zone_a_subnet (10.0.') and '<- admin_network_subnet' in l0[50]
    assert l0[58].startswith('BlockTrafficZone') and l0[65].endswith(')') and 'restricted_zone_b_subnet' in l0[65]
    l4 = env.action_labels('blue_agent_4')
    assert l4[48] == 'Monitor' and l4[145] == 'Sleep' and l4[0].startswith(('Analyse admin', '[Invalid] Analyse admin'))
    ipm, cm = cy.get_ip_map(), cy.get_cidr_map()
    assert len(cm) == 9 and len(set(cm.values())) == 9
    assert 41 <= len(ipm) <= 137 and len(set(ipm.values())) == len(ipm)
    for host, ip in ipm.items():
        sub = [s for s in cm if host.startswith(s)] or ['internet_subnet']
        assert ip.rsplit('.', 1)[0] == cm[sub[0]].rsplit('.', 1)[0]
    assert env.hosts('blue_agent_0')[0] == 'restricted_zone_a_subnet_router'
    assert env.subnets('blue_agent_4') == ['admin_network_subnet', 'office_network_subnet', 'public_access_zone_subnet']


def test_enterprise_wrapper_messages_padding_and_mae(oracle_lib):
    ent = BlueEnterpriseWrapper(CybORG(make_sg(50), seed=9, vec_factory=OracleVecEnv), pad_spaces=True)
    o, i = ent.reset()
    assert all(v.shape == (210,) for v in o.values()) and ent.action_space('blue_agent_0').n == 242
    assert len(i['blue_agent_0']['action_mask']) == 242 and not any(i['blue_agent_0']['action_mask'][82:])
    o, r, te, tr, i = ent.step({'actions': {'blue_agent_0': 241}, 'messages': {'blue_agent_1': np.ones(8, bool)}})
    assert o['blue_agent_0'][60:68].tolist() == [1] * 8 and o['blue_agent_0'][68:92].sum() == 0
    assert o['blue_agent_1'][60:92].sum() == 0
    assert o['blue_agent_4'][178:210].tolist() == [0] * 8 + [1] * 8 + [0] * 16
    mae = EnterpriseMAE(CybORG(make_sg(6), seed=9, vec_factory=OracleVecEnv))
    mae.reset()
    flags = []
    for t in range(5):
        o, r, te, tr, i = mae.step({'blue_agent_0': 16})
        flags.append((te['__all__'], tr['__all__']))
    assert flags == [(False, False)] * 4 + [(False, True)]
    try:
        mae.step({})
        mae.step({})
        raised = False
    except ValueError:
        raised = True                # stepping past `steps` raises like State.py:539-540
    assert raised


class ScriptedAgent:
    def __init__(self, k):
        self.k, self.t = k, 0

    def get_action(self, obs, action_space):
        a = (7 * self.t + 13 * self.k + int(np.asarray(obs).sum())) % action_space.n
        self.t += 1
        return a


def make_submission():
    class Submission:
        NAME, TEAM, TECHNIQUE = 'golden', 'cc4-amd', 'scripted'
        AGENTS = {f'blue_agent_{k}': ScriptedAgent(k) for k in range(5)}

        @staticmethod
        def wrap(env):
            return BlueFlatWrapper(env)
    return Submission


def test_sequential_evaluation_matches_reference_scores(oracle_lib, tmp_path):
    gold = json.load(open(os.path.join(G.GOLDEN_DIR, 'eval_seed321.json')))
    scores = run_evaluation(make_submission(), str(tmp_path), max_eps=gold['episodes'], seed=gold['seed'],
                            mode='sequential', vec_factory=OracleVecEnv)
    assert scores == gold['total_reward']
    assert (tmp_path / 'scores.txt').read_text().startswith('reward_mean: ')
    js = json.load(open(tmp_path / 'summary.json'))
    assert js['parameters'] == {'seed': 321, 'episode_length': 500, 'max_episodes': gold['episodes']}


def test_batched_evaluation_runs_and_is_consistent(oracle_lib, tmp_path):
    class Vec:
        def get_actions(self, obs, mask):
            return np.full(obs.shape[0], 49 if obs.shape[1] == 92 else 145)    # Sleep
    class Sub:
        NAME, TEAM, TECHNIQUE = 'sleep', 't', 'none'
        AGENTS = {f'blue_agent_{k}': Vec() for k in range(5)}
    s = run_evaluation(Sub, str(tmp_path), max_eps=6, seed=1000, mode='batched', episode_length=60, vec_factory=OracleVecEnv)
    assert len(s) == 6 and all(v <= 0 for v in s)
    # episode i of the batch == a single-episode run seeded seed+i (env independence), done-step excluded
    env = OracleVecEnv(1, steps=60)
    env.reset(seeds=np.array([1003], np.uint64))
    tot = 0.0
    for t in range(60):
        _, rew, done, _ = env.step(np.array([[49, 49, 49, 49, 145]], np.int32))
        if done[0]:
            break
        tot += float(rew[0])
    assert s[3] == tot


def test_padded_action_slot_is_an_explicit_sleep_not_a_missing_action(oracle_lib):
    """with blue_agent_class=cc4BlueRandomAgent an agent WITHOUT a submitted action is played by the built-in
    policy; a padded slot ('[Padding] Sleep', index >= 82 with pad_spaces=True) is a submitted Sleep() as in the reference
    (BlueFixedActionWrapper.py:142-148), so the built-in policy must not be asked and draws nothing."""
    from cage_challenge_4_amd import cc4BlueRandomAgent
    sg = EnterpriseScenarioGenerator(blue_agent_class=cc4BlueRandomAgent, green_agent_class=EnterpriseGreenAgent,
                                     red_agent_class=FiniteStateRedAgent, steps=60)
    pad = BlueEnterpriseWrapper(CybORG(sg, seed=77, vec_factory=OracleVecEnv), pad_spaces=True)
    ref = BlueEnterpriseWrapper(CybORG(sg, seed=77, vec_factory=OracleVecEnv), pad_spaces=True)
    pad.reset(); ref.reset()
    sleep = {a: pad.action_labels(a).index('Sleep') for a in pad.possible_agents}
    assert all(pad.action_labels(a)[100] == '[Padding] Sleep' for a in pad.possible_agents[:4])
    for t in range(40):
        a_pad = {a: (100 + t % 50 if b < 4 else sleep[a]) for b, a in enumerate(pad.possible_agents)}
        o1, r1, *_ = pad.step({'actions': a_pad})
        o2, r2, *_ = ref.step({'actions': dict(sleep)})
        assert all(np.array_equal(o1[a], o2[a]) for a in o1) and r1 == r2, t
    assert np.array_equal(pad.env.vec.rng_state(), ref.env.vec.rng_state())      # nothing was drawn for the padded agents
    assert np.array_equal(pad.env.vec.get_state(0), ref.env.vec.get_state(0))
