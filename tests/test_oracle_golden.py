"""CPU: the oracle (host build of the engine restatement) against the golden trajectories recorded from the REAL
reference (oracle/refgen/make_golden.py).  Bit-exact: flat observations, team reward, done, action mask and the
numpy PCG64 stream position after every step."""
import numpy as np
import pytest
import golden_util as G
from oracle_binding import OracleVecEnv


def replay(env_cls, fix, **kw):
    env = env_cls(1, steps=fix['steps'], **kw)
    if fix['reset_seed'] < 0:          # CybORG(seed=s); wrapper.reset()
        env.reset(seeds=np.array([fix['seed']], np.uint64))
        obs = env.reset(seeds=None)
    else:                              # wrapper.reset(seed=s+1)
        env.reset(seeds=np.array([fix['seed']], np.uint64))
        obs = env.reset(seeds=np.array([fix['reset_seed']], np.uint64))
    return env, obs


@pytest.mark.parametrize('path', G.list_fixtures(), ids=lambda p: p.split('/')[-1])
def test_oracle_matches_reference_trajectory(path, oracle_lib):
    fix = G.load(path)
    env, obs = replay(OracleVecEnv, fix, red_policy=fix['red_policy'], green_policy=fix['green_policy'], blue_policy=fix['blue_policy'])
    assert np.array_equal(obs[0], fix['obs'][0])
    assert np.array_equal(env.mask()[0], fix['mask'])
    assert G.rng_words_match(fix['rng'][0], env.rng_state()[0])
    T = fix['actions'].shape[0]
    total = 0.0
    for t in range(T):
        m = None if fix['messages'] is None else fix['messages'][t][None]
        obs, rew, done, info = env.step(fix['actions'][t][None], m)
        assert np.array_equal(obs[0], fix['obs'][t + 1]), f'obs differs at step {t}'
        assert rew[0] == fix['reward'][t], f'reward differs at step {t}'
        assert bool(done[0]) == bool(fix['done'][t]), f'done differs at step {t}'
        assert G.rng_words_match(fix['rng'][t + 1], env.rng_state()[0]), f'PCG64 stream position differs at step {t}'
        assert info['err'][0] == 0
        total += float(rew[0])
    assert total == float(fix['reward'].sum())
    env.close()


def test_baseline_config1_episode_rewards(oracle_lib):
    # seed 123 -> -5634, seed 7 -> -4154 (SleepAgent blue, 500 steps)
    want = {'traj_seed123_sleep_ctor_500.npz': -5634.0, 'traj_seed7_sleep_ctor_500.npz': -4154.0}
    for p in G.list_fixtures():
        name = p.split('/')[-1]
        if name in want:
            assert float(G.load(p)['reward'].sum()) == want[name]
