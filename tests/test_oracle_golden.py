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


def test_oracle_counter_mode_matches_reference_under_the_philox_proxy(oracle_lib):
    """the counter (Philox) mode's STEP pinned to reference trajectories.  tests/golden/ctrstep_*.npz were
    recorded from the real reference stepping under oracle/refgen/philox_proxy.PhiloxProxy -- a Generator stand-in passed as
    CybORG(seed=proxy) that serves every draw of the step path from the engine's (draw, stream, step, episode) Philox words,
    the stream identified from the call site -- on scenarios generated from the numpy stream on both sides.  Every step's
    observations, reward and done flag of ten episodes (random / structured / built-in blue, three red policies, both green
    policies, messages) must match."""
    import ctypes
    from oracle_binding import OracleVecEnv
    allf = [G.load_ctr(p) for p in G.list_ctr_fixtures()]
    assert len(allf) >= 10 and sum(f['steps'] == 500 for f in allf) >= 9 and {f['red_policy'] for f in allf} == {0, 2, 3}
    for steps in sorted({f['steps'] for f in allf}):
        fixes = [f for f in allf if f['steps'] == steps]
        env, obs0, masks = G.ctr_start(OracleVecEnv, fixes)
        for i, f in enumerate(fixes):
            assert np.array_equal(obs0[i], f['obs'][0]) and np.array_equal(masks[i], f['mask']), f['name']
        for t in range(steps):
            for i, f in enumerate(fixes):
                a = np.ascontiguousarray(f['actions'][t], np.int32)
                m = np.ascontiguousarray(f['messages'][t] if f['messages'] is not None else np.zeros((5, 8), np.uint8))
                env.lib.cc4o_step(env._h, i, a.ctypes.data_as(ctypes.c_void_p), m.ctypes.data_as(ctypes.c_void_p))
                env._collect(i)
                assert np.array_equal(env._obs[i], f['obs'][t + 1]), (f['name'], t)
                assert env._rew[i] == f['reward'][t] and bool(env._done[i]) == bool(f['done'][t]), (f['name'], t)
                assert env._err[i] == 0, (f['name'], t)
        env.close()
