"""GPU: the HIP path (through the C ABI of libcc4.so) against (a) the golden trajectories recorded from the real
reference, (b) the CPU oracle on seeded inputs incl. the full packed state, and (c) size-independent properties at
BASELINE.json's full batch sizes.  Bit-exact everywhere: the path is integer / byte / index work."""
import numpy as np
import pytest
import golden_util as G
from oracle_binding import OracleVecEnv, random_actions

pytestmark = pytest.mark.gpu


def _dev(n, **kw):
    from cage_challenge_4_amd import CC4VecEnv
    return CC4VecEnv(n, **kw)


@pytest.fixture(params=[0, 1], ids=['4wave', '1wave'])
def philox_kernel(request, monkeypatch):
    """The counter mode has two step kernels -- four wavefronts per episode (small batches) and one (more than eight episodes
    per CU), picked by cc4_create from the batch size; CC4_PHILOX_LEAN forces one, so that the small test batches cover both."""
    monkeypatch.setenv('CC4_PHILOX_LEAN', str(request.param))
    return ('k_step_philox', 'k_step_philox1')[request.param]


def test_hip_matches_reference_golden_trajectories():
    """All golden episodes stepped together in ONE batch: each env its own seed / init mode / action script."""
    fixes = [G.load(p) for p in G.list_fixtures()]
    fixes = [f for f in fixes if f['steps'] == 500 and f['red_policy'] == 0 and f['green_policy'] == 0 and f['blue_policy'] == 0]
    n = len(fixes)
    env = _dev(n, steps=500)
    env.reset(seeds=np.array([f['seed'] for f in fixes], np.uint64))
    ctor = np.array([f['reset_seed'] < 0 for f in fixes], np.uint8)
    env.reset(seeds=None, env_mask=ctor)                               # CybORG(seed); wrapper.reset()
    second = np.array([max(f['reset_seed'], 0) for f in fixes], np.uint64)
    obs = env.reset(seeds=second, env_mask=1 - ctor)                  # wrapper.reset(seed=s+1)
    for i, f in enumerate(fixes):
        assert np.array_equal(obs[i], f['obs'][0]), f['name']
        assert np.array_equal(env.action_mask[i], f['mask']), f['name']
    T = max(f['actions'].shape[0] for f in fixes)
    for t in range(T):
        a = np.stack([f['actions'][t] for f in fixes])
        m = np.stack([f['messages'][t] if f['messages'] is not None else np.zeros((5, 8), np.uint8) for f in fixes])
        obs, rew, done, info = env.step(a, m)
        st = env.rng_state()
        for i, f in enumerate(fixes):
            assert np.array_equal(obs[i], f['obs'][t + 1]), (f['name'], t)
            assert rew[i] == f['reward'][t], (f['name'], t)
            assert bool(done[i]) == bool(f['done'][t]), (f['name'], t)
            assert G.rng_words_match(f['rng'][t + 1], st[i]), (f['name'], t)
        assert not info['err'].any()
    env.close()


def test_hip_policy_variants_match_reference_golden():
    """DiscoveryFSRed / RandomSelectRedAgent / SleepAgent red / SleepAgent green / cc4BlueRandomAgent blue (acting for agents without a
    submitted action) episodes recorded from the reference, on the HIP path."""
    import test_oracle_golden as T
    from cage_challenge_4_amd import CC4VecEnv
    todo = [f for f in (G.load(p) for p in G.list_fixtures()) if f['red_policy'] or f['green_policy'] or f['blue_policy'] or f['steps'] != 500]
    assert len(todo) >= 7 and sum(f['blue_policy'] for f in todo) >= 3 and sum(f['steps'] == 1000 for f in todo) >= 3   # incl. the 1000-step episodes
    for fix in todo:
        env, obs = T.replay(CC4VecEnv, fix, red_policy=fix['red_policy'], green_policy=fix['green_policy'], blue_policy=fix['blue_policy'])
        assert np.array_equal(obs[0], fix['obs'][0]) and np.array_equal(env.action_mask[0], fix['mask'])
        for t in range(fix['actions'].shape[0]):
            obs, rew, done, info = env.step(fix['actions'][t][None], None if fix['messages'] is None else fix['messages'][t][None])
            assert np.array_equal(obs[0], fix['obs'][t + 1]), (fix['name'], t)
            assert rew[0] == fix['reward'][t] and bool(done[0]) == bool(fix['done'][t]), (fix['name'], t)
            assert G.rng_words_match(fix['rng'][t + 1], env.rng_state()[0]), (fix['name'], t)
        env.close()


def test_hip_counter_mode_matches_reference_under_the_philox_proxy(philox_kernel):
    """VERDICT r02 #2, on the device: both counter-mode step kernels replay the trajectories the real reference produced while
    stepping under oracle/refgen/philox_proxy.PhiloxProxy (tests/golden/ctrstep_*.npz, oracle/refgen/make_ctr_golden.py): the
    scenario from the numpy stream (a numpy-stream handle generates it, its snapshot moves to a counter-mode handle through
    cc4_get/set_state + cc4_get/set_cold), the dynamics on the counter streams of the fixture's key (cc4_set_seed).  All
    episodes of one length in ONE batch; observations, reward, done of every step."""
    from cage_challenge_4_amd import CC4VecEnv
    allf = [G.load_ctr(p) for p in G.list_ctr_fixtures()]
    assert len(allf) >= 10
    for steps in sorted({f['steps'] for f in allf}):
        fixes = [f for f in allf if f['steps'] == steps]
        env, obs0, masks = G.ctr_start(CC4VecEnv, fixes)
        assert env.step_kernel == philox_kernel
        for i, f in enumerate(fixes):
            assert np.array_equal(obs0[i], f['obs'][0]) and np.array_equal(masks[i], f['mask']), f['name']
        for t in range(steps):
            a = np.stack([f['actions'][t] for f in fixes])
            m = np.stack([f['messages'][t] if f['messages'] is not None else np.zeros((5, 8), np.uint8) for f in fixes])
            obs, rew, done, info = env.step(a, m)
            for i, f in enumerate(fixes):
                assert np.array_equal(obs[i], f['obs'][t + 1]), (f['name'], t)
                assert rew[i] == f['reward'][t] and bool(done[i]) == bool(f['done'][t]), (f['name'], t)
            assert not info['err'].any()
        env.close()


@pytest.mark.parametrize('policies', [(0, 0), (2, 0), (1, 1), (3, 0)], ids=['fsm', 'discovery', 'sleep', 'randomselect'])
@pytest.mark.parametrize('rng_mode', [0, 1], ids=['pcg64', 'philox'])
def test_hip_matches_oracle_bit_for_bit(rng_mode, policies, philox_kernel):
    if rng_mode == 0 and philox_kernel != 'k_step_philox':
        pytest.skip('one numpy-stream kernel')
    n, T = 96, 160
    rp, gp = policies
    dev = _dev(n, steps=150, rng_mode=rng_mode, autoreset=True, red_policy=rp, green_policy=gp)
    assert dev.step_kernel == ('k_step' if rng_mode == 0 else philox_kernel)
    ora = OracleVecEnv(n, steps=150, rng_mode=rng_mode, autoreset=True, red_policy=rp, green_policy=gp)
    assert np.array_equal(dev.reset(seeds=31337), ora.reset(seeds=31337))
    assert np.array_equal(dev.action_mask, ora.mask())
    rs = np.random.default_rng(5)
    for t in range(T):                                   # crosses the episode end -> exercises autoreset
        a = random_actions(31337, t, n)
        m = rs.integers(0, 2, size=(n, 5, 8)).astype(np.uint8)
        d = dev.step(a, m)
        o = ora.step(a, m)
        assert np.array_equal(d[0], o[0]), t
        assert np.array_equal(d[1], o[1]), t
        assert np.array_equal(d[2], o[2]), t
        assert np.array_equal(d[3]['err'], o[3]['err']), t
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(0, n, 7):
        a_, b_ = dev.get_state(i), ora.get_state(i)
        assert np.array_equal(a_, b_), f'packed state differs env {i} at byte offsets {np.nonzero(a_ != b_)[0][:20].tolist()}'
    dev.close()


@pytest.mark.parametrize('rng_mode', [0, 1], ids=['pcg64', 'philox'])
def test_builtin_blue_policy_matches_oracle(rng_mode, philox_kernel):
    """blue_agent_class=cc4BlueRandomAgent: the built-in policy acts for every blue agent without a submitted
    action -- here a changing subset -- and draws from the episode's stream (its own counter streams in the Philox mode)."""
    if rng_mode == 0 and philox_kernel != 'k_step_philox':
        pytest.skip('one numpy-stream kernel')
    n, T = 96, 170
    dev = _dev(n, steps=150, rng_mode=rng_mode, autoreset=True, red_policy=3, blue_policy=1)
    ora = OracleVecEnv(n, steps=150, rng_mode=rng_mode, autoreset=True, red_policy=3, blue_policy=1)
    assert np.array_equal(dev.reset(seeds=4242), ora.reset(seeds=4242))
    for t in range(T):
        a = random_actions(4242, t, n)
        idle = (np.arange(n)[:, None] + np.arange(5)[None, :] + t) % 3 != 0      # two thirds of the agents leave it to the policy
        a[idle] = -1
        if t % 7 == 0:
            a[:] = -1
        d = dev.step(a); o = ora.step(a)
        assert np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1]) and np.array_equal(d[2], o[2]), t
        assert np.array_equal(d[3]['err'], o[3]['err']), t
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(0, n, 5):
        assert np.array_equal(dev.get_state(i), ora.get_state(i)), i
    dev.close()


@pytest.mark.parametrize('rng_mode', [0, 1], ids=['pcg64', 'philox'])
def test_full_batch_matches_oracle_every_step(rng_mode, philox_kernel):
    """BASELINE configs[1] size: all 1024 episodes against the oracle at every step (observations, rewards, dones, error
    flags), across an episode end (autoreset), and the packed state of every episode at the end.  A batch this size meets
    the rare interleavings (same-step phishing + reassignment, concurrent blue Restore / red exploit on neighbouring
    tables) that the 96-episode matrix can miss."""
    if rng_mode == 0 and philox_kernel != 'k_step_philox':
        pytest.skip('one numpy-stream kernel')
    n, T = 1024, 260
    dev = _dev(n, steps=220, rng_mode=rng_mode, autoreset=True)
    assert dev.step_kernel == ('k_step' if rng_mode == 0 else philox_kernel)
    ora = OracleVecEnv(n, steps=220, rng_mode=rng_mode, autoreset=True)
    assert np.array_equal(dev.reset(seeds=777), ora.reset(seeds=777))
    for t in range(T):
        a = random_actions(777, t, n)
        d = dev.step(a)
        o = ora.step(a)
        bad = np.nonzero((d[0] != o[0]).any(axis=1) | (d[1] != o[1]) | (d[2] != o[2]))[0]
        assert bad.size == 0, (t, bad[:10].tolist())
        assert not d[3]['err'].any(), t
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(n):
        a_, b_ = dev.get_state(i), ora.get_state(i)
        assert np.array_equal(a_, b_), f'packed state differs env {i} at byte offsets {np.nonzero(a_ != b_)[0][:20].tolist()}'
    dev.close()


@pytest.mark.parametrize('n,red_policy,blue_policy,T,steps,rng_mode', [(8192, 0, 0, 1100, 500, 1), (8192, 2, 0, 330, 150, 1), (8192, 3, 1, 330, 150, 1), (4096, 0, 0, 330, 150, 1),
                                                                       (8192, 0, 0, 1100, 500, 0)],
                         ids=['8192-fsm-500-step-episodes', '8192-discovery', '8192-randomselect-builtinblue', '4096-fsm', '8192-fsm-500-step-episodes-numpy-stream'])
def test_bench_configuration_matches_oracle_every_step(n, red_policy, blue_policy, T, steps, rng_mode):
    """the configuration bench.py times -- counter mode, autoreset, the kernel cc4_create picks for the batch
    size with NO override (at 8192 episodes k_step_philox1 at its natural residency -- generation work area in HBM, host rows in
    L2 with atomics -- as three concurrent launches on three streams) -- against the oracle for ALL episodes at EVERY step (observations, rewards, dones, error flags)
    across two scenario regenerations, then the generator words and the packed state of every episode."""
    import os
    assert 'CC4_PHILOX_LEAN' not in os.environ and 'CC4_PHILOX_MINW' not in os.environ and 'CC4_GROUPS' not in os.environ
    # (two regenerations each; the headline configuration -- 8192 episodes, FiniteStateRedAgent, 500-step episodes: BASELINE configs[2] as written --
    # for 1100 steps, the other red policies and the built-in blue policy with 150-step episodes; r05 ran these at 75 / 50-step episodes while the
    # oracle's OpenMP pool oversubscribed the box's cgroup -- tests/oracle_binding.usable_cores.  Since r05 a host-stepped batch is ONE launch per step on
    # the main stream (the caller fetches every step: launch_step); the launches per episode group run in test_timed_bench_path_matches_oracle's
    # one-step bursts and in test_learner_loop_matches_oracle_in_every_launch_form)
    dev = _dev(n, steps=steps, rng_mode=rng_mode, autoreset=True, red_policy=red_policy, blue_policy=blue_policy)
    # three or four launches of the one-wave kernel per step (four where the runtime runs four streams side by side); the last case: the same
    # batch on the numpy stream (bench.py's alt_rng: bit-exact with the reference itself), kernel k_step
    assert dev.step_kernel == ('k_step_philox1' if rng_mode else 'k_step') and dev.lib.cc4_launches_per_step(dev._h) in (3, 4)
    ora = OracleVecEnv(n, steps=steps, rng_mode=rng_mode, autoreset=True, red_policy=red_policy, blue_policy=blue_policy)
    assert np.array_equal(dev.reset(seeds=1000), ora.reset_batch(1000))
    resets = 0
    for t in range(T):
        a = random_actions(1000, t, n)
        if blue_policy:
            a[(np.arange(n)[:, None] + np.arange(5)[None, :] + t) % 3 == 0] = -1      # a third of the agents leave it to the built-in policy
        resets += int(ora._done.all())
        d = dev.step(a)
        o = ora.step_batch(a)
        bad = np.nonzero((d[0] != o[0]).any(axis=1) | (d[1] != o[1]) | (d[2] != o[2]) | (d[3]['err'] != o[3]['err']))[0]
        assert bad.size == 0, (t, bad[:10].tolist())
        assert not d[3]['err'].any(), t
    assert resets == 2
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(n):
        a_, b_ = dev.get_state(i), ora.get_state(i)
        assert np.array_equal(a_, b_), f'packed state differs env {i} at byte offsets {np.nonzero(a_ != b_)[0][:20].tolist()}'
    dev.close()


@pytest.mark.parametrize('n,kernel,run_kernel', [(8192, 'k_step_philox1', 'k_run_philox1'), (1024, 'k_step_philox', 'k_run_philox'), (2048, 'k_step_philox', 'k_run_philox8'),
                                                  (4096, 'k_step_philox1', 'k_run_philox1m'), (16384, 'k_step_philox1', 'k_run_philox1')],
                         ids=['8192', '1024-multistep', '2048-multistep8', '4096-multistep1', '16384-partitions-of-64'])
def test_timed_bench_path_matches_oracle(n, kernel, run_kernel):
    """the exact region bench.py times -- cc4_run_random_steps on k_step_philox1, 8192 episodes, the handle's own
    launch grouping, no override, no communicator: the blue actions are drawn IN the step kernel on the bank lanes (BK_BRAND) --
    against the oracle driven with the host restatement of the same draws (random_actions), in bursts of K = 1, 20 and 100 steps
    (the driver's --steps 20 among them) across a scenario regeneration: observations, rewards, dones, error flags and the
    contents of the device action buffer (the actions of the burst's last step) after every burst, then generator words and the
    packed state of all episodes."""
    import ctypes, os
    assert 'CC4_PHILOX_LEAN' not in os.environ and 'CC4_PHILOX_MINW' not in os.environ and 'CC4_GROUPS' not in os.environ
    steps, seed0 = (150, 4242) if n == 8192 else (100, 4242)
    bursts = (1, 20, 137, 20, 1, 20, 9, 64) if n == 8192 else (1, 20, 57, 12, 3, 40)
    dev = _dev(n, steps=steps, rng_mode=1, autoreset=True)
    # (1024 episodes = BASELINE configs[1]: the chip holds the batch at once, and the region is ONE launch of the multi-step kernel
    # k_run_philox, every block looping over the steps of its episode)
    assert dev.step_kernel == kernel and dev.run_kernel == run_kernel and dev.launches_per_step in (3, 4)
    ora = OracleVecEnv(n, steps=steps, rng_mode=1, autoreset=True)
    assert np.array_equal(dev.reset(seeds=seed0), ora.reset_batch(seed0))
    t = 0
    for K in bursts:
        dev.run_random_steps(seed0, t, K, timed=(K != 1))
        for k in range(K):
            a = random_actions(seed0, t + k, n)
            o = ora.step_batch(a)
        t += K
        dev.synchronize()
        dev._fetch()
        bad = np.nonzero((dev._obs != o[0]).any(axis=1) | (dev._rew != o[1]) | (dev._done.astype(bool) != o[2]) | (dev._err != o[3]['err']))[0]
        assert bad.size == 0, (K, t, bad[:10].tolist())
        assert np.array_equal(dev.device_actions(), a), (K, t)
    assert t > steps                                     # every episode was regenerated once inside a timed burst
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(n):
        a_, b_ = dev.get_state(i), ora.get_state(i)
        assert np.array_equal(a_, b_), f'packed state differs env {i} at byte offsets {np.nonzero(a_ != b_)[0][:20].tolist()}'
    dev.close()


@pytest.mark.parametrize('threads', ['0', '1'])
def test_enqueue_threads_change_nothing(threads, monkeypatch):
    """cc4_run_random_steps with one enqueue thread per group stream (the default where the host has eight hardware threads)
    and from the calling thread alone: bursts of 1 / 12 / 3 / 14 steps across a regeneration against the oracle, handles created and
    destroyed in a row (a worker pool is joined at cc4_destroy)."""
    monkeypatch.setenv('CC4_ENQ_THREADS', threads)
    monkeypatch.setenv('CC4_RUN1', '0')
    monkeypatch.setenv('CC4_PERSIST', '0')
    n, steps, seed0 = 4096, 25, 99
    for rep in range(3):
        dev = _dev(n, steps=steps, rng_mode=1, autoreset=True)
        assert dev.launches_per_step in (3, 4) and dev.run_kernel == 'k_step_philox1'      # (CC4_RUN1=0 below: the per-step launches are what the threads serve)
        ora = OracleVecEnv(n, steps=steps, rng_mode=1, autoreset=True)
        assert np.array_equal(dev.reset(seeds=seed0 + rep), ora.reset_batch(seed0 + rep))
        t = 0
        for K in (1, 20, 3, 57):
            dev.run_random_steps(seed0, t, K, timed=(K != 3))
            for k in range(K):
                o = ora.step_batch(random_actions(seed0, t + k, n))
            t += K
            dev.synchronize(); dev._fetch()
            assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1]) and np.array_equal(dev._done.astype(bool), o[2]), (rep, K)
        d = dev.step(random_actions(seed0, t, n)); o = ora.step_batch(random_actions(seed0, t, n))     # a plain step behind the joined end
        assert np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1])
        assert np.array_equal(dev.rng_state(), ora.rng_state())
        dev.close(); ora.close()


@pytest.mark.parametrize('mode,kernel', [(1, 'k_run_philox1'), (0, 'k_run_pcg')], ids=['counter', 'numpy-stream'])
def test_persistent_kernel_and_its_shared_tail(mode, kernel):
    """The persistent run kernel of large batches: many short calls -- every call ends in a tail whose items the CUs of an XCD share
    behind an agent-scope acquire -- of varying length across a regeneration against the oracle; a batch just beyond what one launch holds
    (partitions of 20-26 episodes for 20-24 waves) and calls too short for it (per-step launches) in between."""
    n, steps, seed0 = (6656 if mode else 5000), 60, 31337
    dev = _dev(n, steps=steps, rng_mode=mode, autoreset=True)
    assert dev.run_kernel == kernel and dev.run_kernel_for(9) == dev.step_kernel and dev.run_kernel_for(10) == kernel
    ora = OracleVecEnv(n, steps=steps, rng_mode=mode, autoreset=True)
    assert np.array_equal(dev.reset(seeds=seed0), ora.reset_batch(seed0))
    t = 0
    for K in (10, 13, 3, 10, 25, 11, 40, 10, 1, 17):
        dev.run_random_steps(seed0, t, K, timed=(K % 2 == 1))
        for k in range(K):
            a = random_actions(seed0, t + k, n)
            o = ora.step_batch(a)
        t += K
        dev.synchronize(); dev._fetch()
        bad = np.nonzero((dev._obs != o[0]).any(axis=1) | (dev._rew != o[1]) | (dev._done.astype(bool) != o[2]) | (dev._err != o[3]['err']))[0]
        assert bad.size == 0, (K, t, bad[:10].tolist())
        assert np.array_equal(dev.device_actions(), a), (K, t)
    assert t > steps
    for k in range(3):                                   # plain steps with the caller's actions behind the one-launch calls, then another call
        a = random_actions(seed0 + 1, t + k, n)
        d = dev.step(a); o = ora.step_batch(a)
        assert np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1]) and np.array_equal(d[2], o[2]), k
    t += 3
    dev.run_random_steps(seed0, t, 12, timed=True)
    for k in range(12):
        o = ora.step_batch(random_actions(seed0, t + k, n))
    dev.synchronize(); dev._fetch()
    assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1])
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(0, n, 3):
        assert np.array_equal(dev.get_state(i), ora.get_state(i)), i
    dev.close(); ora.close()


@pytest.mark.gpu
@pytest.mark.parametrize('n', [6656, 8192])
def test_short_persistent_calls_leave_every_hot_row_equal_to_the_oracle(n):
    """the ordering bug r05 shipped for a day (a stale hot row after a short call, one call in sixty) passed every parity test and was
    found only by the opt-in self-check.  This one runs in the DEFAULT configuration: many short calls -- each ends in a tail the CUs of an XCD share,
    each hands every episode from wave to wave a few times -- and after every call the full packed state of EVERY episode (hot rows in one fetch, cold
    rows sampled) against the oracle, not only the outputs."""
    import ctypes
    steps, seed0 = 90, 4711 + n
    dev = _dev(n, steps=steps, rng_mode=1, autoreset=True)
    assert dev.run_kernel_for(10) == 'k_run_philox1'
    ora = OracleVecEnv(n, steps=steps, rng_mode=1, autoreset=True)
    assert np.array_equal(dev.reset(seeds=seed0), ora.reset_batch(seed0))
    nb = int(dev.lib.cc4_state_bytes())
    t = 0
    for c, K in enumerate((10, 12, 10, 11, 13, 10, 20, 10, 15, 10, 11, 10, 17, 10, 12, 10, 14, 10, 10, 21, 10, 13, 10, 16) * 2):
        dev.run_random_steps(seed0, t, K, timed=False)
        for k in range(K):
            o = ora.step_batch(random_actions(seed0, t + k, n))
        t += K
        dev.synchronize(); dev._fetch()
        assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1]), (c, K, t)
        rows = dev.get_states()                                         # [n][state bytes], one copy
        want = np.stack([ora.get_state(i) for i in range(n)]) if c % 6 == 5 else None
        if want is not None:
            bad = np.nonzero((rows != want).any(axis=1))[0]
            assert bad.size == 0, (c, K, t, bad[:8].tolist())
        else:                                                           # the other calls: a third of the rows
            for i in range(c % 3, n, 3):
                assert np.array_equal(rows[i], ora.get_state(i)), (c, K, t, i)
        for i in range(c % 97, n, 97):
            assert np.array_equal(dev.get_cold(i), ora.get_cold(i)), (c, K, t, i)
    assert rows.shape == (n, nb)
    dev.close(); ora.close()


@pytest.mark.gpu
def test_sampled_self_check_runs_by_default(monkeypatch):
    """without CC4_PERSIST_VERIFY every CC4_PERSIST_VERIFY_EVERY-th persistent call (default 1024) is repeated on a shadow handle
    and compared all the same; here every 5th, and the counters of cc4_verify_stats say so."""
    monkeypatch.delenv('CC4_PERSIST_VERIFY', raising=False)
    monkeypatch.setenv('CC4_PERSIST_VERIFY_EVERY', '5')
    dev = _dev(8192, steps=100, rng_mode=1, autoreset=True, strict=False); dev.reset(seeds=5)
    t = 0
    for c in range(16):
        dev.run_random_steps(5, t, 10 + c % 3, timed=False); t += 10 + c % 3
    assert dev.verify_stats() == (3, 0)
    dev.run_random_steps(5, t, 4, timed=False)              # (too short for the persistent form: not counted)
    assert dev.verify_stats() == (3, 0)
    dev.close()


def _pack_rows(obs):
    """[n, 578] observation values (0 / 1 / 2) -> [n, 148] bytes, 2 bits per value, low bits first (CC4_OBS_PACKED_BYTES)."""
    n = obs.shape[0]
    v = np.zeros((n, 592), np.uint8)
    v[:, :578] = obs.astype(np.uint8) & 3
    v = v.reshape(n, 148, 4)
    return (v[:, :, 0] | (v[:, :, 1] << 2) | (v[:, :, 2] << 4) | (v[:, :, 3] << 6)).astype(np.uint8)


def _hash_policy(packed, j):
    """numpy restatement of k_rollout_hash_policy (csrc/cc4_k_misc.hip): FNV-1a over the 37 words of an episode's packed observation row."""
    w = np.ascontiguousarray(packed).view('<u4').astype(np.uint64)          # [n, 37]
    h = np.full(w.shape[0], 2166136261, np.uint64)
    for c in range(w.shape[1]):
        h = ((h ^ w[:, c]) * np.uint64(16777619)) & np.uint64(0xFFFFFFFF)
    out = np.zeros((w.shape[0], 5), np.int32)
    for b in range(5):
        out[:, b] = ((h + np.uint64(2654435761 * (b + 1)) + np.uint64(40503 * j)) & np.uint64(0xFFFFFFFF)) % np.uint64(242 if b == 4 else 82)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize('n,policy', [(8192, 'random'), (6656, 'hash'), (8192, 'hash')])
def test_rollout_with_the_policy_in_the_loop_matches_the_oracle_at_every_step(n, policy):
    """policy -> step -> policy (CybORG/Evaluation/evaluation.py:89-110) with ONE launch of the step engine per rollout: the persistent
    kernel waits, per step and policy group, for the actions the caller's stream publishes; the policy reads the packed observations of the step
    before behind a gate.  Checked at EVERY step: the packed observation rows the policy read, the actions the steps consumed (the 'hash' stand-in
    computes them FROM those rows -- a stale or missing row changes the trajectory), and at the end of each rollout outputs, generator positions and
    packed states against the oracle.  Rollouts of up to 32 steps (the ring of observation slabs), across a regeneration."""
    steps, seed0 = 60, 777 + n
    dev = _dev(n, steps=steps, rng_mode=1, autoreset=True)
    ora = OracleVecEnv(n, steps=steps, rng_mode=1, autoreset=True)
    o_prev = ora.reset_batch(seed0).copy()
    assert np.array_equal(dev.reset(seeds=seed0), o_prev)
    assert dev.run_kernel_for(20) == 'k_run_philox1'
    t = 0
    for c, K in enumerate((12, 25, 31, 10)):
        dev.run_rollout(K, policy, seed0, t, native=(c % 2 == 1))        # (the passes enqueued from Python / by cc4_rollout_standin)
        acts = []
        for j in range(K):
            a = random_actions(seed0, t + j, n) if policy == 'random' else _hash_policy(_pack_rows(o_prev), j)
            assert np.array_equal(dev.rollout_obs_packed(j), _pack_rows(o_prev)), (K, j, 'the rows the policy of this step read')
            o = ora.step_batch(a)
            o_prev = o[0].copy()
            acts.append(a)
        t += K
        dev._fetch()
        assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1]) and np.array_equal(dev._done.astype(bool), o[2]), K
        assert np.array_equal(dev.rollout_actions(K - 1), acts[-1]) and np.array_equal(dev.rollout_actions(K - 2), acts[-2]), K
        assert not dev._err.any()
    assert t > steps and np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(0, n, 7):
        assert np.array_equal(dev.get_state(i), ora.get_state(i)), i
    # behind the rollouts: plain steps and a one-launch call still find the handle where the steps left it
    a = random_actions(seed0 + 1, t, n)
    d = dev.step(a); o = ora.step_batch(a)
    assert np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1])
    dev.close(); ora.close()


@pytest.mark.gpu
def test_rollout_watchdog_reports_a_pass_that_was_never_published(monkeypatch):
    """A rollout whose policy never publishes must not hang the kernel: the waiting steps give up after the watchdog, cc4_rollout_end says so (-6)."""
    monkeypatch.setenv('CC4_ROLLOUT_WATCHDOG_MS', '20')
    import ctypes
    dev = _dev(8192, steps=50, rng_mode=1, autoreset=True, strict=False); dev.reset(seeds=3)
    dev._chk(dev.lib.cc4_rollout_begin(dev._h, 10), 'cc4_rollout_begin')
    for g in range(4):                                       # step 0 only
        dev.lib.cc4_rollout_wait_obs(dev._h, g, 0, None); dev.lib.cc4_rollout_random_policy(dev._h, g, 0, 3, 0, None); dev.lib.cc4_rollout_publish(dev._h, g, 0, None)
    obs = np.zeros((8192, 578), np.int32)
    assert dev.lib.cc4_get_obs(dev._h, obs.ctypes.data_as(ctypes.c_void_p)) != 0 and b'a rollout is in flight' in dev.lib.cc4_last_error(dev._h)     # the handle is the rollout's alone
    assert dev.lib.cc4_synchronize(dev._h) != 0 and dev.lib.cc4_rollout_begin(dev._h, 5) != 0
    assert dev.lib.cc4_rollout_end(dev._h) == -6
    assert b'waited longer than' in dev.lib.cc4_last_error(dev._h)
    assert dev.lib.cc4_get_obs(dev._h, obs.ctypes.data_as(ctypes.c_void_p)) == 0
    dev.close()


def test_device_random_action_kernel_matches_host_restatement():
    import ctypes
    n = 1024
    dev = _dev(n, steps=20)
    dev.reset(seeds=1)
    dev._chk(dev.lib.cc4_random_actions_device(dev._h, ctypes.c_uint64(1000), 7), 'rand')
    dev.synchronize()
    # read back through a step on the device buffer: compare with stepping the host-generated actions
    ora = OracleVecEnv(n, steps=20); ora.reset(seeds=1)
    p = ctypes.c_void_p()
    dev.lib.cc4_actions_device(dev._h, ctypes.byref(p))
    dev._chk(dev.lib.cc4_step_device(dev._h, p, None), 'step_device')
    dev.synchronize()
    obs, rew, done = dev._fetch()
    o = ora.step(random_actions(1000, 7, n))
    assert np.array_equal(obs, o[0]) and np.array_equal(rew, o[1])
    dev.close()


@pytest.mark.parametrize('n', [8192, 1500])
def test_learner_loop_matches_oracle_in_every_launch_form(n, monkeypatch):
    """The loop a learner runs -- a kernel writes the actions on the device, cc4_step_device consumes them, no host wait in between -- in the
    three forms the library has for it: a policy over the WHOLE batch (the library then steps the batch with ONE launch on the main stream),
    the same with CC4_WHOLE_BATCH_STEPS=0 (a launch per episode group, forked from and joined to the main stream every step), and the policy
    applied per episode group on the group's own stream.  Every form against the oracle at every step, the generator position at the end."""
    T, seed0 = 8, 4321
    ora = OracleVecEnv(n, steps=40, rng_mode=1); ora.reset(seeds=77)
    want = []
    for t in range(T):
        o = ora.step_batch(random_actions(seed0, t, n))
        want.append((o[0].copy(), o[1].copy(), np.asarray(o[2]).astype(bool).copy()))
    for t in range(T, T + 5):
        ora.step_batch(random_actions(seed0, t, n))
    rng_end = ora.rng_state().copy()
    ora.close()
    for form in ('whole', 'per-group launches', 'grouped policy'):
        monkeypatch.setenv('CC4_WHOLE_BATCH_STEPS', '0' if form == 'per-group launches' else '1')
        dev = _dev(n, steps=40, rng_mode=1); dev.reset(seeds=77)
        for t in range(T):
            (dev.run_policy_steps_grouped if form == 'grouped policy' else dev.run_policy_steps)(seed0, t, 1)
            dev.synchronize(); dev._fetch()
            o = want[t]
            bad = np.nonzero((dev._obs != o[0]).any(axis=1) | (dev._rew != o[1]) | (dev._done.astype(bool) != o[2]))[0]
            assert bad.size == 0, (form, t, bad[:10].tolist())
        dev.run_policy_steps(seed0, T, 5)                       # five more without a look in between, then the generator position
        dev.synchronize()
        assert np.array_equal(dev.rng_state(), rng_end), form
        dev.close()


@pytest.mark.parametrize('rng_mode', [0, 1], ids=['pcg64', 'philox'])
@pytest.mark.parametrize('n', [1024, 8192])
def test_full_size_properties(n, rng_mode):
    """BASELINE configs 2/3, both RNG modes (the kernel is the one cc4_create picks for the batch size: at 8192 episodes in the
    counter mode the one-wave kernel, checked here against a six-episode batch on the four-wave kernel and against the
    oracle): determinism, batch-independence of every episode, observation invariants, clean flags."""
    T = 40
    a_env = _dev(n, steps=500, rng_mode=rng_mode); b_env = _dev(n, steps=500, rng_mode=rng_mode)
    if rng_mode == 1:
        assert a_env.step_kernel == ('k_step_philox1' if n > 2048 else 'k_step_philox')
    oa = a_env.reset(seeds=1000).copy(); ob = b_env.reset(seeds=1000).copy()
    assert np.array_equal(oa, ob)
    small_idx = np.array([0, 1, n // 3, n // 2, n - 2, n - 1])
    small = _dev(len(small_idx), steps=500, rng_mode=rng_mode)
    if rng_mode == 1:
        assert small.step_kernel == 'k_step_philox'
    small.reset(seeds=np.uint64(1000) + small_idx.astype(np.uint64))
    ora = OracleVecEnv(len(small_idx), steps=500, rng_mode=rng_mode)
    ora.reset(seeds=np.uint64(1000) + small_idx.astype(np.uint64))
    digest_a = digest_b = 0
    for t in range(T):
        acts = random_actions(1000, t, n)
        oa, ra, da, ia = a_env.step(acts); ob, rb, db, ib = b_env.step(acts)
        assert np.array_equal(oa, ob) and np.array_equal(ra, rb)            # run-to-run determinism
        os_, rs_, *_ = small.step(acts[small_idx])
        assert np.array_equal(oa[small_idx], os_) and np.array_equal(ra[small_idx], rs_)   # batch independence
        oo, ro, *_ = ora.step(acts[small_idx])
        assert np.array_equal(os_, oo) and np.array_equal(rs_, ro)                          # == oracle
        assert not ia['err'].any()
        v = oa
        for off, ns in ((0, 1), (92, 1), (184, 1), (276, 1), (368, 3)):
            assert (v[:, off] == v[:, 0]).all() and ((v[:, off] >= 0) & (v[:, off] <= 2)).all()
            for k in range(ns):
                blk = v[:, off + 1 + 59 * k: off + 1 + 59 * (k + 1)]
                assert (blk[:, 0:9].sum(1) == 1).all()
        body = np.delete(v, [0, 92, 184, 276, 368], axis=1)
        assert ((body == 0) | (body == 1)).all()
        assert (ra <= 0).all()
    assert np.array_equal(a_env.rng_state(), b_env.rng_state())
    for e in (a_env, b_env, small):
        e.close()


def test_rccl_allgather_world_size_one():
    """The native RCCL exchange step (cc4_comm_init / cc4_allgather_obs) on a 1-rank communicator: gathered == local."""
    import ctypes, os
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    n = 256
    dev = _dev(n, steps=50); dev.reset(seeds=3)
    ident = (ctypes.c_uint8 * 128)()
    assert dev.lib.cc4_comm_unique_id(ident) == 0
    dev._chk(dev.lib.cc4_comm_init(dev._h, 0, 1, ident), 'cc4_comm_init')
    from cage_challenge_4_amd import distributed as D
    ora = OracleVecEnv(n, steps=50); ora.reset(seeds=3)
    for t in range(4):
        obs, *_ = dev.step(random_actions(3, t, n))
        o = ora.step(random_actions(3, t, n))
        assert D.allgather_obs_device(dev)
        D.allgather_wait(dev)
        got = D.allgathered_obs_host(dev, 1)
        assert np.array_equal(got, obs.astype(np.uint8)) and np.array_equal(obs, o[0])
    # the timed bench loop (in-kernel random actions, overlapped all-gather on the comm stream, double-buffered bytes)
    dev.run_random_steps(3, 4, 7, timed=True)
    for t in range(4, 11):
        o = ora.step(random_actions(3, t, n))
    dev._fetch()
    assert np.array_equal(dev._obs, o[0])
    assert np.array_equal(D.allgathered_obs_host(dev, 1), o[0].astype(np.uint8))
    dev.close()


def _pack(obs):
    """[M, 578] values 0..2 -> [M, 148] bytes: value i in bits 2 * (i & 3) of byte i >> 2 (include/cc4.h CC4_OBS_PACKED_BYTES)."""
    v = np.zeros((obs.shape[0], 592), np.uint8)
    v[:, :578] = obs
    v = v.reshape(obs.shape[0], 148, 4)
    return (v[:, :, 0] | (v[:, :, 1] << 2) | (v[:, :, 2] << 4) | (v[:, :, 3] << 6)).astype(np.uint8)


def _one_rank_comm(dev):
    import ctypes, os
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    ident = (ctypes.c_uint8 * 128)()
    assert dev.lib.cc4_comm_unique_id(ident) == 0
    dev._chk(dev.lib.cc4_comm_init(dev._h, 0, 1, ident), 'cc4_comm_init')


@pytest.mark.parametrize('n,groups', [(768, '3')])
def test_observation_ring_survives_a_slow_exchange(n, groups, monkeypatch):
    """VERDICT r02 #8c, the per-step exchange (CC4_EXCHANGE_INKERNEL=0: what a handle falls back to): the step kernels write their packed
    observations into a ring of 8 buffers that overlapped all-gathers read; a host-side guard keeps a launch from overwriting a buffer whose
    all-gather has not finished.  With cc4_debug_comm_delay_us every all-gather is preceded by ~150 us of idling on the communication
    stream -- several steps' worth -- so the guard has to wait again and again (cc4_host_stats counts it), and nothing may be lost."""
    monkeypatch.setenv('CC4_GROUPS', groups)
    monkeypatch.setenv('CC4_EXCHANGE_INKERNEL', '0')
    from cage_challenge_4_amd import distributed as D
    dev = _dev(n, steps=120, rng_mode=1, autoreset=True); dev.reset(seeds=77)
    assert dev.launches_per_step == int(groups)
    _one_rank_comm(dev)
    assert dev.launches_per_step == int(groups) and not dev.exchange_info()['in_kernel']      # CC4_GROUPS pins the grouping through cc4_comm_init
    dev.reset(seeds=77)
    dev._chk(dev.lib.cc4_debug_comm_delay_us(dev._h, 150), 'cc4_debug_comm_delay_us')
    ora = OracleVecEnv(n, steps=120, rng_mode=1, autoreset=True); ora.reset_batch(77)
    t = 0
    for burst in (1, 3, 9, 17, 40, 33):
        dev.run_random_steps(77, t, burst, timed=(burst % 2 == 1))          # step + all-gather per step, nothing waits in between
        for k in range(burst):
            o = ora.step_batch(random_actions(77, t + k, n))
        t += burst
        dev._fetch()
        assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1]), burst
        assert np.array_equal(D.allgathered_obs_host(dev, 1), o[0].astype(np.uint8)), burst      # the LAST step's gather, intact
    st = dev.host_stats()
    assert st['gathers'] == t and st['gather_stalls'] >= 10, st      # the guard really had to wait for the slow exchange
    dev._chk(dev.lib.cc4_debug_comm_delay_us(dev._h, 0), 'cc4_debug_comm_delay_us')
    dev.close()


@pytest.mark.parametrize('n,mode,run_kernel,delay_us', [(1000, 1, 'k_run_philox', 1500), (2000, 1, 'k_run_philox1m', 0),
                                                        (8192, 1, 'k_run_philox1x', 0), (6656, 1, 'k_run_philox1x', 2500), (5000, 0, 'k_run_pcg', 0)],
                         ids=['1000-slow-exchange', '2000', '8192', '6656-slow-exchange', '5000-numpy-stream'])     # (1000 / 2000: the last group of 32 episodes is a partial one)
def test_exchange_from_inside_the_one_launch_kernels_gathers_every_step(n, mode, run_kernel, delay_us):
    """with a communicator cc4_run_random_steps stays ONE launch -- step k writes its packed rows into slab k mod 32 of a
    ring and counts finished episodes, the communication stream waits for the count (hipStreamWaitValue32), all-gathers the slab and
    publishes how far it got (hipStreamWriteValue32) in chunks of 8 steps, step k + 32 waits for that.  Checked here on a one-rank communicator: the gathered
    rows of EVERY step (cc4_debug_gather_log), not only the last of a burst, equal the oracle's -- also with an exchange several times
    slower than the steps, which makes the kernel wait for its slabs --, then observations, rewards, generator words and packed state."""
    from cage_challenge_4_amd import distributed as D
    steps, seed0 = 60, 2024
    dev = _dev(n, steps=steps, rng_mode=mode, autoreset=True); dev.reset(seeds=seed0)
    _one_rank_comm(dev)
    xi = dev.exchange_info()
    assert xi['in_kernel'] and xi['ring'] == 32 and xi['chunk'] == 8 and dev.run_kernel_for(20) == run_kernel, (xi, dev.run_kernel_for(20))
    dev.reset(seeds=seed0)
    if delay_us:
        dev._chk(dev.lib.cc4_debug_comm_delay_us(dev._h, delay_us), 'cc4_debug_comm_delay_us')
    ora = OracleVecEnv(n, steps=steps, rng_mode=mode, autoreset=True); ora.reset_batch(seed0)
    t = 0
    for K in (20, 40, 12):                                                   # across a regeneration; a burst longer than the ring
        dev.gather_log(K)
        dev.run_random_steps(seed0, t, K, timed=True)
        want = []
        for k in range(K):
            o = ora.step_batch(random_actions(seed0, t + k, n))
            want.append(_pack(o[0].astype(np.uint8)))
        got = dev.get_gather_log(1, 0, K)
        for k in range(K):
            bad = np.nonzero((got[k] != want[k]).any(axis=1))[0]
            assert bad.size == 0, (K, t + k, bad[:10].tolist())
        t += K
        dev._fetch()
        assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1]) and np.array_equal(dev._err, o[3]['err']), K
        assert np.array_equal(D.allgathered_obs_host(dev, 1), o[0].astype(np.uint8)), K
    xi = dev.exchange_info()
    assert xi['calls'] == 3 and xi['watchdog_timeouts'] == 0 and xi['in_kernel'], xi
    # an explicit all-gather right behind a one-launch call: the per-step ring's buffer is filled from the exchange ring's last slab on demand
    assert D.allgather_obs_device(dev); D.allgather_wait(dev)
    assert np.array_equal(D.allgathered_obs_host(dev, 1), o[0].astype(np.uint8))
    # a per-step launch with the caller's actions and an explicit all-gather behind the one-launch calls
    a = random_actions(seed0 + 1, t, n)
    d = dev.step(a); o = ora.step_batch(a)
    assert np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1])
    assert D.allgather_obs_device(dev); D.allgather_wait(dev)
    assert np.array_equal(D.allgathered_obs_host(dev, 1), o[0].astype(np.uint8))
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(0, n, 5):
        assert np.array_equal(dev.get_state(i), ora.get_state(i)), i
    dev.close(); ora.close()


def test_exchange_watchdog_returns_the_handle_to_per_step_launches(monkeypatch):
    """An exchange that never catches up -- every all-gather held back by 30 ms, watchdog 5 ms -- must not hang the one-launch kernel: the
    waiting step gives up, the call completes with every episode intact, stderr says so, and the handle runs per-step launches from then on."""
    monkeypatch.setenv('CC4_EXCHANGE_WATCHDOG_MS', '5')
    from cage_challenge_4_amd import distributed as D
    n, seed0 = 512, 7
    dev = _dev(n, steps=200, rng_mode=1, autoreset=True); dev.reset(seeds=seed0)
    _one_rank_comm(dev); dev.reset(seeds=seed0)
    assert dev.exchange_info()['in_kernel'] and dev.run_kernel_for(40) == 'k_run_philox'
    dev._chk(dev.lib.cc4_debug_comm_delay_us(dev._h, 30000), 'cc4_debug_comm_delay_us')
    ora = OracleVecEnv(n, steps=200, rng_mode=1, autoreset=True); ora.reset_batch(seed0)
    dev.run_random_steps(seed0, 0, 40, timed=False)
    for k in range(40):
        o = ora.step_batch(random_actions(seed0, k, n))
    dev._fetch()
    assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1])            # the episodes never depended on the exchange
    xi = dev.exchange_info()
    assert xi['watchdog_timeouts'] == 1 and not xi['in_kernel'] and dev.run_kernel_for(40) == dev.step_kernel, xi
    dev._chk(dev.lib.cc4_debug_comm_delay_us(dev._h, 0), 'cc4_debug_comm_delay_us')
    dev.run_random_steps(seed0, 40, 5, timed=False)                                      # per-step launches + per-step all-gathers now
    for k in range(40, 45):
        o = ora.step_batch(random_actions(seed0, k, n))
    dev._fetch()
    assert np.array_equal(dev._obs, o[0]) and np.array_equal(D.allgathered_obs_host(dev, 1), o[0].astype(np.uint8))
    dev.close(); ora.close()


@pytest.mark.parametrize('n,mode', [(4096, 1), (8192, 1), (5000, 0)], ids=['4096-multistep1', '8192-persistent', '5000-numpy-stream'])
def test_exchange_soak_random_call_lengths_and_delays(n, mode):
    """The exchange from inside the one-launch kernels under a random schedule (tests/soak_exchange.py, shortened): calls of 10..90 steps -- every other one
    longer than the ring of 32 slabs --, every all-gather held up by a random 0..60 us, across regenerations; every step's gathered rows against the oracle."""
    seed0, steps = 2468, 120
    dev = _dev(n, steps=steps, rng_mode=mode, autoreset=True); dev.reset(seeds=seed0)
    _one_rank_comm(dev)
    ora = OracleVecEnv(n, steps=steps, rng_mode=mode, autoreset=True); ora.reset_batch(seed0)
    rng = np.random.default_rng(n + mode)
    t = 0
    for c in range(8):
        K = int(rng.integers(10, 91))
        dev._chk(dev.lib.cc4_debug_comm_delay_us(dev._h, int(rng.integers(0, 61))), 'cc4_debug_comm_delay_us')
        dev.gather_log(K)
        dev.run_random_steps(seed0, t, K, timed=False)
        got = dev.get_gather_log(1, 0, K)
        for k in range(K):
            o = ora.step_batch(random_actions(seed0, t + k, n))
            bad = np.nonzero((got[k] != _pack(o[0].astype(np.uint8))).any(axis=1))[0]
            assert bad.size == 0, (c, K, t + k, bad[:10].tolist())
        t += K
    xi = dev.exchange_info()
    assert xi['in_kernel'] and xi['calls'] == 8 and xi['watchdog_timeouts'] == 0, xi
    assert np.array_equal(dev.rng_state(), ora.rng_state())
    for i in range(0, n, 7):
        assert np.array_equal(dev.get_state(i), ora.get_state(i)), i
    dev.close(); ora.close()


@pytest.mark.parametrize('n', [8192, 6656])
def test_persistent_kernel_self_check_mode(n, monkeypatch):
    """CC4_PERSIST_VERIFY=1: the persistent kernel hands an episode from one wave to the next of the same CU with
    workgroup-scope ordering only (agent-scope ordering costs 40 % -- profiles/r05_persist_order_ab.txt); in this mode the library
    repeats every one-launch call with per-step launches on a shadow handle and compares hot rows, cold rows and outputs of all episodes.
    6656 episodes: partitions of 26 for 24 waves (r05: 5632, 22 for 20) -- every call ends in a tail shared across the CUs of an XCD.  (It has earned its keep: r05
    replaced the hand-over's explicit s_waitcnt vmcnt(0) by a workgroup-scope release fence -- for which the backend emits no vmcnt wait
    without tgsplit -- and this test caught the stale row, one call in about sixty.)"""
    monkeypatch.setenv('CC4_PERSIST_VERIFY', '1')
    dev = _dev(n, steps=100, rng_mode=1, autoreset=True, strict=False); dev.reset(seeds=99)
    assert dev.run_kernel_for(20) == 'k_run_philox1'
    t = 0
    calls = (20, 10, 37, 20, 64, 10, 13, 11, 25, 10, 17, 12) * (3 if n == 6656 else 1)    # short calls: each one ends in a shared tail
    for K in calls:
        dev.run_random_steps(99, t, K, timed=(K == 20)); t += K
    assert dev.verify_stats() == (len(calls), 0)
    dev.run_random_steps(99, t, 5, timed=False)                                           # too short for the one-launch form: nothing to verify
    assert dev.verify_stats() == (len(calls), 0)
    dev.close()


def test_two_real_ranks_gather_the_unsharded_batch(tmp_path):
    """Two processes on two devices, a real two-rank RCCL communicator (the first time two GPUs are visible: the 1-GPU boxes of the pool
    skip): every step's gathered rows == the oracle's unsharded batch, from inside the one-launch kernel and from per-step launches;
    cc4_comm_info reports two ranks on two devices."""
    import os, subprocess, sys
    from cage_challenge_4_amd import _lib
    if int(_lib.load().cc4_device_count()) < 2:
        pytest.skip('needs two GPUs')
    from conftest import ROOT
    worker = os.path.join(ROOT, 'tests', '_rccl_worker.py')
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29611', NCCL_SOCKET_IFNAME='lo',
                   HSA_ENABLE_IPC_MODE_LEGACY='0', CC4_CONTROL_PLANE_KEY=f'rccl2_{os.getpid()}', CC4_CONTROL_PLANE_DIR=str(tmp_path))
        env.pop('CC4_EXCHANGE_INKERNEL', None)
        procs.append(subprocess.Popen([sys.executable, worker], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outs = [p.communicate(timeout=1500) for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[0][-2500:] + o[1][-2500:] for o in outs)
    assert 'RCCL_OK 2' in outs[0][0]


def test_snapshot_restore_replays_identically(philox_kernel):
    """SURVEY 8(f)-4 (snapshot tooling): state + cold row of one episode captured, the episode advanced, restored and advanced
    again -> identical trajectory; the other episodes of the batch are unaffected."""
    dev = _dev(4, steps=100); dev.reset(seeds=9)
    for t in range(25):
        dev.step(random_actions(9, t, 4))
    snap = dev.snapshot(2)
    first = []
    for t in range(25, 45):
        o, r, d, _ = dev.step(random_actions(9, t, 4))
        first.append((o.copy(), r.copy()))
    rng_a = dev.rng_state()[2].copy()
    dev.restore(2, snap)
    for k, t in enumerate(range(25, 45)):
        o, r, d, _ = dev.step(random_actions(9, t, 4))
        assert np.array_equal(o[2], first[k][0][2]) and r[2] == first[k][1][2]
    assert np.array_equal(dev.rng_state()[2], rng_a)
    assert len(snap[0]) == dev.lib.cc4_state_bytes() and len(snap[1]) == dev.lib.cc4_cold_bytes(dev._h)
    dev.close()


def test_measurement_hook_ends_a_step_behind_a_phase_and_restores_cleanly(monkeypatch):
    """cc4_debug_stop_phase (tools/valu_phases.py: per-phase instruction counts): phase 14 = a whole step of the full build of k_step_philox1 == the
    oracle's step; a step ended behind an earlier phase writes no row back (the batch, restored, continues exactly where it stood); 0 = off."""
    monkeypatch.setenv('CC4_PHILOX_LEAN', '1')
    for k in ('CC4_PERSIST', 'CC4_MULTISTEP', 'CC4_RUN1'):
        monkeypatch.setenv(k, '0')
    n = 96
    dev = _dev(n, steps=60, rng_mode=1, autoreset=True); ora = OracleVecEnv(n, steps=60, rng_mode=1, autoreset=True)
    assert np.array_equal(dev.reset(seeds=4400), ora.reset_batch(4400)) and dev.step_kernel == 'k_step_philox1'
    t = 0
    for _ in range(70):                                       # across a regeneration
        dev.run_random_steps(4400, t, 1, timed=False); ora.step_batch(random_actions(4400, t, n)); t += 1
    snaps = [dev.snapshot(i) for i in range(n)]
    for stop in (3, 7, 9, 12):
        assert dev.lib.cc4_debug_stop_phase(dev._h, stop) == 0
        dev.run_random_steps(4400, t, 1, timed=False); dev.synchronize()
        for i in range(n):
            dev.restore(i, snaps[i])
    assert dev.lib.cc4_debug_stop_phase(dev._h, 14) == 0
    dev.run_random_steps(4400, t, 1, timed=False); dev.synchronize(); dev._fetch()
    o = ora.step_batch(random_actions(4400, t, n)); t += 1
    assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev._rew, o[1]) and np.array_equal(dev._done.astype(bool), o[2])
    assert dev.lib.cc4_debug_stop_phase(dev._h, 0) == 0 and dev.lib.cc4_debug_stop_phase(dev._h, 15) != 0
    for _ in range(20):
        dev.run_random_steps(4400, t, 1, timed=False); o = ora.step_batch(random_actions(4400, t, n)); t += 1
    dev.synchronize(); dev._fetch()
    assert np.array_equal(dev._obs, o[0]) and np.array_equal(dev.rng_state(), ora.rng_state())
    assert all(np.array_equal(dev.get_state(i), ora.get_state(i)) for i in range(n))
    dev.close(); ora.close()


def test_drop_in_wrapper_reproduces_reference_episode():
    """The mirror of BlueFlatWrapper(CybORG(EnterpriseScenarioGenerator(...), seed=123)) replays the golden episode."""
    from cage_challenge_4_amd import (CybORG, EnterpriseScenarioGenerator, SleepAgent, EnterpriseGreenAgent,
                                      FiniteStateRedAgent, BlueFlatWrapper, EnterpriseMAE)
    fix = G.load([p for p in G.list_fixtures() if 'seed123_random_ctor_500' in p][0])
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent,
                                     red_agent_class=FiniteStateRedAgent, steps=500)
    env = BlueFlatWrapper(CybORG(sg, seed=123))
    obs, info = env.reset()
    assert env.agents == [f'blue_agent_{b}' for b in range(5)]
    assert [env.action_space(a).n for a in env.agents] == [82, 82, 82, 82, 242]
    assert [len(env.observation_space(a)) for a in env.agents] == [92, 92, 92, 92, 210]
    flat = np.concatenate([obs[a] for a in env.agents])
    assert flat.dtype == np.int64 and np.array_equal(flat, fix['obs'][0])
    assert np.array_equal(np.concatenate([info[a]['action_mask'] for a in env.agents]), fix['mask'])
    labels = env.action_labels('blue_agent_0')
    assert labels[16] == 'Monitor' and labels[49] == 'Sleep' and labels[0].endswith('restricted_zone_a_subnet_server_host_0')
    total = 0.0
    for t in range(120):
        acts = {f'blue_agent_{b}': int(fix['actions'][t, b]) for b in range(5)}
        obs, rew, term, trunc, info = env.step(acts)
        assert np.array_equal(np.concatenate([obs[a] for a in env.possible_agents]), fix['obs'][t + 1]), t
        assert set(rew.values()) == {float(fix['reward'][t])}
        assert term['blue_agent_0'] == bool(fix['done'][t]) == trunc['blue_agent_0']
        total += rew['blue_agent_0']
    env.close()
    from cage_challenge_4_amd import BlueEnterpriseWrapper
    ent = BlueEnterpriseWrapper(CybORG(sg, seed=123), pad_spaces=True)
    o, i = ent.reset()
    assert all(v.shape == (210,) for v in o.values()) and ent.action_space('blue_agent_0').n == 242
    o, r, te, tr, i = ent.step({'actions': {'blue_agent_0': 241}, 'messages': {'blue_agent_1': np.ones(8, bool)}})
    assert o['blue_agent_0'][92 - 32: 92 - 24].tolist() == [1] * 8      # agent 1's message is agent 0's first message slot
    assert o['blue_agent_1'][92 - 32:92].sum() == 0                      # nobody else sent anything
    ent.close()
    mae = EnterpriseMAE(CybORG(sg, seed=123))                            # RLlib flavour: plain dict + "__all__" keys
    mae.reset()
    o, r, te, tr, i = mae.step({'blue_agent_0': 16})
    assert te['__all__'] is False and tr['__all__'] is False and set(o) == set(mae.possible_agents)
    mae.close()


def test_evaluation_harness_on_device():
    """the batched evaluation loop on the HIP backend: sequential mode reproduces the reference's scores,
    batched mode agrees with single-episode runs of the same seeds."""
    import json, os
    from cage_challenge_4_amd.evaluation import run_evaluation
    from test_wrappers_cpu import make_submission
    gold = json.load(open(os.path.join(G.GOLDEN_DIR, 'eval_seed321.json')))
    scores = run_evaluation(make_submission(), None, max_eps=2, seed=gold['seed'], mode='sequential', write_to_file=False)
    assert scores == gold['total_reward'][:2]

    class Vec:
        def get_actions(self, obs, mask):
            return np.full(obs.shape[0], 49 if obs.shape[1] == 92 else 145)
    class Sub:
        NAME, TEAM, TECHNIQUE = 'sleep', 't', 'none'
        AGENTS = {f'blue_agent_{k}': Vec() for k in range(5)}
    s = run_evaluation(Sub, None, max_eps=100, seed=1000, mode='batched', episode_length=80, write_to_file=False)
    ora = OracleVecEnv(100, steps=80); ora.reset(seeds=1000)
    tot = np.zeros(100); alive = np.ones(100, bool)
    acts = np.tile(np.array([[49, 49, 49, 49, 145]], np.int32), (100, 1))
    for t in range(80):
        _, rew, done, _ = ora.step(acts)
        alive &= ~done
        tot += np.where(alive, rew, 0.0)
    assert s == [float(v) for v in tot]


def _episode_statistics(mode, n, T, seed0):
    """One batch of n full T-step episodes with device-side random blue actions; per-episode statistics that every draw site
    of the transition feeds: rewards per mission phase, event-flag counts seen by the blue agents (connection events: scans,
    decoy alerts, blocked green traffic; process events: exploits, green false positives), red session counts at three times,
    Impact executions, suspicious-pid totals."""
    import ctypes
    from oracle_binding import load as load_oracle
    lay = ctypes.create_string_buffer(8192)
    load_oracle().cc4o_layout(lay, 8192)
    off = {ln.split()[0]: int(ln.split()[1]) for ln in lay.value.decode().splitlines() if len(ln.split()) == 2}
    RA, BA = off['sizeof.RedAgent'], off['sizeof.BlueAgent']
    env = _dev(n, steps=T, rng_mode=mode)
    env.reset(seeds=seed0)
    st = {k: np.zeros(n) for k in ('reward', 'rew_p0', 'rew_p1', 'rew_p2', 'conn_flags', 'proc_flags', 'impacts',
                                   'sess_t1', 'sess_t2', 'sess_end', 'sus_end')}
    third = T // 3
    conn = np.zeros(578, bool); proc = np.zeros(578, bool)
    for base in (0, 92, 184, 276):
        proc[base + 28:base + 44] = True; conn[base + 44:base + 60] = True
    for i in range(3):
        proc[368 + 28 + 59 * i:368 + 44 + 59 * i] = True; conn[368 + 44 + 59 * i:368 + 60 + 59 * i] = True

    def red_fields():
        ns = np.zeros(n); imp = np.zeros(n); sus = np.zeros(n)
        for e in range(n):
            row = env.get_state(e)
            for r in range(6):
                o = off['red'] + r * RA
                ns[e] += row[o + off['red.nsess']]
                imp[e] += (row[o + off['red.exec_type']] == 6) and row[o + off['red.nsess']] > 0
            for b in range(5):
                o = off['blue'] + b * BA
                sus[e] += int(row[o + 32]) | (int(row[o + 33]) << 8)
        return ns, imp, sus
    for t in range(T - 1):
        env.run_random_steps(555, t, 1, timed=False)
        obs, rew, done = env._fetch()
        st['reward'] += rew
        st['rew_p%d' % min(2, t // third)] += rew
        st['conn_flags'] += obs[:, conn].sum(1)
        st['proc_flags'] += obs[:, proc].sum(1)
        if t in (T // 5, (3 * T) // 5, T - 2):
            ns, imp, sus = red_fields()
            st[{T // 5: 'sess_t1', (3 * T) // 5: 'sess_t2', T - 2: 'sess_end'}[t]] = ns
            if t == T - 2:
                st['sus_end'] = sus
    assert not env.err.any()
    # Impact executions: every executed Impact of a red agent that still holds a session costs the RIA entry of the
    # BlueRewardMachine; counted through the reward it leaves (exact multiples are not needed for a distribution test)
    st.pop('impacts')
    env.close()
    return st


def test_philox_and_pcg_modes_agree_in_distribution_on_device():
    """VERDICT r01 #6 (the counter mode has no reference trajectory to replay, so it is pinned in distribution): 16384 full
    300-step episodes per RNG mode on the HIP path, random blue actions from the same device generator; two-sample
    Kolmogorov-Smirnov on ten per-episode statistics that between them see every draw site (rewards per mission phase,
    connection / process event flags, red session counts at three times, suspicious pids).  A draw-site bug confined to one
    mode that shifts any of these by a tenth of a standard deviation fails the gate (D ~ 0.04 > 0.03)."""
    from scipy.stats import ks_2samp
    n, T = 16384, 300
    a = _episode_statistics(0, n, T, 70_000)
    b = _episode_statistics(1, n, T, 70_000)
    worst = {}
    for k in a:
        d = ks_2samp(a[k], b[k]).statistic
        worst[k] = (round(float(d), 4), round(float(a[k].mean()), 3), round(float(b[k].mean()), 3))
    assert a['reward'].mean() < -500 and a['sess_end'].mean() > 3 and a['conn_flags'].mean() > 50, worst
    assert all(v[0] < 0.03 for v in worst.values()), worst


def test_shared_topology_mode_matches_oracle():
    """cc4_config.topology_seed: all episodes share the scenario (also after autoreset), dynamics differ, == oracle."""
    n, T = 64, 70
    dev = _dev(n, steps=50, rng_mode=1, autoreset=True, topology_seed=4242)
    ora = OracleVecEnv(n, steps=50, rng_mode=1, autoreset=True, topology_seed=4242)
    assert np.array_equal(dev.reset(seeds=9000), ora.reset(seeds=9000))
    assert len({dev.topology(i).tobytes() for i in range(n)}) == 1
    rew = []
    for t in range(T):
        a = random_actions(9000, t, n)
        d = dev.step(a); o = ora.step(a)
        assert np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1]) and np.array_equal(d[2], o[2]), t
        rew.append(d[1].copy())
    assert len({dev.topology(i).tobytes() for i in range(n)}) == 1          # the regenerated scenario is shared too
    assert (np.array(rew).std(axis=1) > 0).any()                            # ... while the dynamics are per episode
    for i in range(0, n, 9):
        assert np.array_equal(dev.get_state(i), ora.get_state(i))
    with pytest.raises(Exception):
        _dev(4, steps=50, rng_mode=0, topology_seed=1)                    # the numpy stream cannot split scenario from dynamics
    dev.close()


@pytest.mark.parametrize('rng_mode,red_policy', [(1, 0), (0, 0), (1, 2), (1, 3)], ids=['philox-fsm', 'pcg64-fsm', 'philox-discovery', 'philox-randomselect'])
def test_soak_bounded_containers_never_overflow(rng_mode, red_policy):
    """4096 episodes x 1500 steps (three full 500-step episodes each, autoreset) of device-side random blue actions: no
    engine error flag (container overflow, reference-crash path) may ever be raised, rewards stay <= 0, observations stay
    in range, and every episode finishes exactly on the autoreset cadence."""
    n = 4096
    dev = _dev(n, steps=500, rng_mode=rng_mode, autoreset=True, red_policy=red_policy)
    dev.reset(seeds=20000)
    done_count = np.zeros(n, np.int64)
    for chunk in range(15):
        dev.run_random_steps(20000, chunk * 100, 100, timed=False)
        obs, rew, done = dev._fetch()
        assert not dev.err.any(), (chunk, np.unique(dev.err[dev.err != 0])[:5])
        assert (rew <= 0).all() and ((obs >= 0) & (obs <= 2)).all()
        done_count += done
    assert done_count.sum() == 0 or (done_count == done_count[0]).all()      # lock-step episodes: all done flags coincide
    dev.close()


@pytest.mark.gpu
def test_c_abi_error_behaviour():
    """Bad arguments fail loudly with a message (include/cc4.h: negative return + cc4_last_error), never with a crash or a
    silent fallback: invalid configurations, an out-of-range device or episode index, a step past the last mission phase
    (the reference raises ValueError there, State.py:539-540)."""
    import ctypes
    from cage_challenge_4_amd import CC4VecEnv, _lib as L
    for kw in (dict(num_envs=0), dict(num_envs=4, steps=0), dict(num_envs=4, rng_mode=7), dict(num_envs=4, red_policy=9),
               dict(num_envs=4, device_id=99), dict(num_envs=4, topology_seed=5, rng_mode=0)):
        with pytest.raises(L.CC4Error):
            CC4VecEnv(**kw)
    env = CC4VecEnv(4, steps=6, rng_mode=1)
    env.reset(seeds=3)
    buf = np.zeros(env.lib.cc4_state_bytes(), np.uint8)
    assert env.lib.cc4_get_state(env._h, 4, buf.ctypes.data_as(ctypes.c_void_p)) != 0
    assert env.lib.cc4_get_state(env._h, -1, buf.ctypes.data_as(ctypes.c_void_p)) != 0
    assert b'range' in env.lib.cc4_last_error(env._h)
    assert int(env.lib.cc4_get_true_state(env._h, 9, None, 0)) < 0
    for _ in range(5):                                   # steps 0..4 of a 6-step episode are the whole episode
        env.step(np.full((4, 5), -1, np.int32))
    assert env._done.all()
    with pytest.raises(ValueError):                      # no autoreset: stepping on raises like the reference
        for _ in range(3):
            env.step(np.full((4, 5), -1, np.int32))
    env.reset(seeds=3)                                   # and the handle is still usable
    assert not env.step(np.full((4, 5), -1, np.int32))[2].any()


@pytest.mark.gpu
@pytest.mark.parametrize('rng_mode', [0, 1])
def test_masked_reset_touches_only_the_masked_episodes(rng_mode):
    """cc4_reset(seeds, env_mask): the masked episodes become what a fresh handle generates from the same seeds (in the
    counter mode that is the per-host-phase generation of k_reset), the others keep every byte of their state."""
    from cage_challenge_4_amd import CC4VecEnv
    from oracle_binding import random_actions
    n = 24
    env = CC4VecEnv(n, steps=60, rng_mode=rng_mode)
    env.reset(seeds=900)
    for t in range(9):
        env.step(random_actions(77, t, n))
    before = [env.get_state(i).copy() for i in range(n)]
    mask = (np.arange(n) % 3 == 0).astype(np.uint8)
    seeds = np.uint64(5000) + np.arange(n, dtype=np.uint64)
    obs = env.reset(seeds=seeds, env_mask=mask).copy()
    fresh = CC4VecEnv(n, steps=60, rng_mode=rng_mode)
    obs_fresh = fresh.reset(seeds=seeds)
    for i in range(n):
        if mask[i]:
            assert np.array_equal(env.get_state(i), fresh.get_state(i)), i
            assert np.array_equal(obs[i], obs_fresh[i]), i
        else:
            assert np.array_equal(env.get_state(i), before[i]), i
    a = random_actions(78, 0, n)
    o1, r1, d1, _ = env.step(a)
    o2, r2, d2, _ = fresh.step(a)
    sel = mask.astype(bool)
    assert np.array_equal(o1[sel], o2[sel]) and np.array_equal(r1[sel], r2[sel])
    assert not env.err.any()


def _kind_actions(mask, kind, rng, one_host=False):
    """[N,5] wrapper indices: every agent plays `kind` ('decoy' | 'restore' | 'remove') on a random valid host of its zone
    (one_host: always its first valid one).  mask: [N,570] action mask of the batch."""
    n = mask.shape[0]
    a = np.zeros((n, 5), np.int32)
    for b in range(5):
        nsub = 3 if b == 4 else 1
        nh, nc = 16 * nsub, 8 * nsub
        base = {'remove': nh + 1, 'restore': 2 * nh + 1, 'decoy': 3 * nh + 2 + 2 * nc}[kind]
        m = mask[:, (82 * b if b < 4 else 328) + base:(82 * b if b < 4 else 328) + base + nh]
        if one_host:
            a[:, b] = base + m.argmax(axis=1)
        else:
            r = rng.random((n, nh)) * m                       # a random valid slot per episode
            a[:, b] = base + r.argmax(axis=1)
    return a


@pytest.mark.parametrize('rng_mode', [0, 1], ids=['pcg64', 'philox'])
def test_soak_decoy_only_blue_never_overflows(rng_mode):
    """4096 episodes x 500 steps with every blue agent deploying decoys (random valid host; one fixed host for
    the first 256 episodes, which stacks ~250 decoys on it): process lists grow past the hot row into EnvCold.povf, no
    engine error flag is ever raised, and the first episodes stay bit-identical to the CPU oracle including the packed state."""
    n, k = 4096, 8
    dev = _dev(n, steps=500, rng_mode=rng_mode); dev.reset(seeds=31000)
    ora = OracleVecEnv(k, steps=500, rng_mode=rng_mode); ora.reset(seeds=31000)
    mask = dev.action_mask
    rng = np.random.default_rng(5)
    fixed = np.arange(n) < 256
    for t in range(499):
        a = np.where(fixed[:, None], _kind_actions(mask, 'decoy', rng, one_host=True), _kind_actions(mask, 'decoy', rng))
        obs, rew, done, info = dev.step(a)                    # strict mode: raises on any error flag
        o = ora.step(a[:k])
        assert np.array_equal(obs[:k], o[0]) and np.array_equal(rew[:k], o[1]), t
    assert not dev.err.any() and not ora._err.any()
    for i in range(k):
        assert np.array_equal(dev.get_state(i), ora.get_state(i)), i
    import json
    nproc = max(len(h['procs']) for h in json.loads(dev.true_state_json(0))['hosts'])
    assert nproc >= 240, nproc                                # the stack really went far beyond the 8 hot-row slots
    dev.close()


@pytest.mark.parametrize('kind', ['restore', 'remove'])
def test_structured_blue_policies_match_oracle(kind, philox_kernel):
    """Restore-only / Remove-only blue (both RNG modes): HIP == oracle every step and in the packed state."""
    n = 64
    for rng_mode in (0, 1):
        dev = _dev(n, steps=300, rng_mode=rng_mode); dev.reset(seeds=32000)
        ora = OracleVecEnv(n, steps=300, rng_mode=rng_mode); ora.reset(seeds=32000)
        mask = dev.action_mask
        rng = np.random.default_rng(6)
        for t in range(299):
            a = _kind_actions(mask, kind, rng)
            obs, rew, done, info = dev.step(a)
            o = ora.step(a)
            assert np.array_equal(obs, o[0]) and np.array_equal(rew, o[1]), (rng_mode, t)
        for i in range(n):
            assert np.array_equal(dev.get_state(i), ora.get_state(i)), (rng_mode, i)
        dev.close()


def test_exchange_row_of_a_reset_and_device_unpack(philox_kernel):
    """cc4_allgather_obs right after cc4_reset gathers the RESET observations (k_reset writes the packed exchange
    row); cc4_get_allgathered_obs reads the buffer of the last all-gather, not of the last step; cc4_unpack_obs_device turns
    the gathered 2-bit rows into [world*N, 578] bytes on the device."""
    import ctypes, os
    os.environ.setdefault('NCCL_SOCKET_IFNAME', 'lo')
    from cage_challenge_4_amd import distributed as D
    n = 192
    dev = _dev(n, steps=50, rng_mode=1)
    dev.reset(seeds=9)
    out = np.zeros((n, 578), np.uint8)
    assert dev.lib.cc4_get_allgathered_obs(dev._h, out.ctypes.data_as(ctypes.c_void_p)) != 0      # no communicator yet
    ident = (ctypes.c_uint8 * 128)()
    assert dev.lib.cc4_comm_unique_id(ident) == 0
    dev._chk(dev.lib.cc4_comm_init(dev._h, 0, 1, ident), 'cc4_comm_init')
    assert dev.lib.cc4_get_allgathered_obs(dev._h, out.ctypes.data_as(ctypes.c_void_p)) != 0      # nothing gathered yet
    obs0 = dev.reset(seeds=9).copy()
    assert D.allgather_obs_device(dev)
    D.allgather_wait(dev)
    assert np.array_equal(D.allgathered_obs_host(dev, 1), obs0.astype(np.uint8))
    for t in range(3):
        obs, *_ = dev.step(random_actions(9, t, n))
        obs = obs.copy()
        assert D.allgather_obs_device(dev)
        p = ctypes.c_void_p()
        dev._chk(dev.lib.cc4_unpack_obs_device(dev._h, ctypes.byref(p)), 'cc4_unpack_obs_device')
        assert p.value
        D.allgather_wait(dev)
        dev._chk(dev.lib.cc4_get_unpacked_obs(dev._h, out.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_unpacked_obs')
        assert np.array_equal(out, obs.astype(np.uint8)), t
    newer, *_ = dev.step(random_actions(9, 3, n))             # a step after the last gather: the gathered copy must not move
    assert np.array_equal(D.allgathered_obs_host(dev, 1), obs.astype(np.uint8))
    assert not np.array_equal(newer.astype(np.uint8), obs.astype(np.uint8))
    dev.close()


def test_set_seed_matches_oracle():
    """cc4_set_seed (CybORG.set_seed): HIP == oracle after a mid-episode reseed, both RNG modes, incl. the packed state."""
    n = 32
    for rng_mode in (0, 1):
        dev = _dev(n, steps=100, rng_mode=rng_mode); dev.reset(seeds=400)
        ora = OracleVecEnv(n, steps=100, rng_mode=rng_mode); ora.reset(seeds=400)
        for t in range(60):
            if t in (20, 35):
                dev.set_seed(7000 + t); ora.set_seed(7000 + t)
            a = random_actions(400, t, n)
            d, o = dev.step(a), ora.step(a)
            assert np.array_equal(d[0], o[0]) and np.array_equal(d[1], o[1]), (rng_mode, t)
        for i in range(n):
            assert np.array_equal(dev.get_state(i), ora.get_state(i)), (rng_mode, i)
        dev.close()


def test_bench_two_ranks_on_one_gpu_shards_and_reports():
    """the whole N>1 path of bench.py (rendezvous, bench.plan_shard, barrier + max-over-ranks timing, rank-0
    output) with two ranks on the one GPU of the test box.  RCCL refuses two ranks on one device, so the exchange falls back
    and the line must say so; the sharded batch must end exactly where the unsharded one does (episodes are keyed by their
    global index)."""
    import json, os, socket, subprocess, sys
    from conftest import ROOT
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    common = ['--total-envs', '512', '--steps', '20', '--warmup', '5', '--min-seconds', '0', '--no-alt', '--no-cpu-baseline']
    env = dict(os.environ, CC4_BENCH_DEVICE='0', CC4_RCCL_SETUP_TIMEOUT='90', OMP_NUM_THREADS='1', NCCL_SOCKET_IFNAME='lo',
               HSA_ENABLE_IPC_MODE_LEGACY='0')
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2'] + common,
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert two.returncode == 0, two.stdout[-1500:] + two.stderr[-3000:]
    lines2 = [ln for ln in two.stdout.splitlines() if ln.startswith('{')]
    assert len(lines2) == 1, two.stdout[-2000:]                  # ONE JSON line, from rank 0
    d2 = json.loads(lines2[0])
    # world 1 driving the same distributed code path; the distributed run steps once more while it sets the exchange up,
    # then resets, so both runs end after warmup + K steps
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1'] + common, capture_output=True, text=True,
                         timeout=600, env=dict(env, CC4_BENCH_FORCE_DIST='1', CC4_RCCL_SETUP_FAIL='1', MASTER_PORT=str(port + 1) if port < 65000 else '29611'), cwd=ROOT)
    assert one.returncode == 0, one.stdout[-1500:] + one.stderr[-3000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith('{')][0])
    assert d2['n_gpus'] == 2 and d2['scaling'] == 'strong'
    assert d2['config']['total_envs'] == 512 and d2['config']['envs_per_gpu'] == 256 and d1['config']['envs_per_gpu'] == 512
    assert d2['config']['exchange'].startswith(('none (RCCL setup failed', 'RCCL all-gather')), d2['config']['exchange']
    assert d1['config']['exchange'].startswith('none (RCCL setup failed')
    assert not d2['config']['engine_error_flags'] and not d1['config']['engine_error_flags']
    assert d2['config']['steps_run'] == d1['config']['steps_run'] == 25
    assert d2['config']['last_step_reward_sum'] == d1['config']['last_step_reward_sum']
    assert d2['config']['last_step_done_count'] == d1['config']['last_step_done_count']
    assert d2['value'] > 0 and d2['roofline']['launch_ms'] > 0


def test_counter_mode_event_log_ports_do_not_change_the_trajectory(philox_kernel):
    """rng_mode 1 with cc4_enable_event_log: the ephemeral ports of the logged events come from side streams, so (a) the episode
    is the same with and without the log, (b) HIP and oracle write the same event records, (c) the ports are really drawn."""
    import json
    n, T = 8, 120
    plain = _dev(n, steps=200, rng_mode=1); plain.reset(seeds=515)
    logged = _dev(n, steps=200, rng_mode=1); logged.enable_event_log(True); logged.reset(seeds=515)
    ora = OracleVecEnv(n, steps=200, rng_mode=1); ora.enable_event_log(True); ora.reset(seeds=515)
    ports = set()
    for t in range(T):
        a = random_actions(515, t, n)
        p, l, o = plain.step(a), logged.step(a), ora.step(a)
        assert np.array_equal(p[0], l[0]) and np.array_equal(p[1], l[1]), t
        assert np.array_equal(l[0], o[0]) and np.array_equal(l[1], o[1]), t
        if t % 10 == 9:
            for i in range(n):
                dj, oj = json.loads(logged.true_state_json(i)), json.loads(ora.true_state_json(i))
                key = lambda ev: sorted(tuple(e[:1] + e[2:]) for e in ev)      # noqa: E731  (the order of records of different agents is not fixed)
                assert key(dj['events']) == key(oj['events']), (t, i)
                ports.update(e[5] for e in dj['events'] if e[5] >= 49152)
                ports.update(e[7] for e in dj['events'] if e[7] >= 49152)
    assert len(ports) > 5 and 49152 not in ports or len(ports) > 6, sorted(ports)[:10]
    for i in range(n):
        assert np.array_equal(plain.get_state(i), logged.get_state(i))
    plain.close(); logged.close()


def test_bench_line_contract():
    """`python bench.py --steps 20 --warmup 5` (what the driver runs, with its own K / W): ONE JSON line with the contract's keys,
    BASELINE.json's metric on the 8192-episode configuration, `roofline` and `cpu_baseline` objects, consistent numbers."""
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5', '--min-seconds', '0.05'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data',
              'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 20 and d['warmup'] == 5 and d['higher_is_better'] is True and d['vs_baseline'] is None
    assert d['scaling'] == 'strong' and d['data'] == 'synthetic' and d['dtype'] == 'int32' and 'model' not in d['config']
    assert '8192 vectorised envs' in d['config']['workload'] and d['config']['total_envs'] == 8192
    assert not d['config']['engine_error_flags'] and d['config']['timed_regions'] >= 1
    r = d['roofline']
    # `bound` is what the counters say (r06): "hbm" only when the committed counter passes of THIS kernel show the memory system busy, "valu_issue" when
    # they show its vector issue slots busy (the 8192-episode one-launch kernel: nine cycles in ten), else "latency"
    assert r['bound'] in ('hbm', 'valu_issue', 'latency') and r['roofline'] == 'hbm' and r['bound_evidence']
    if r['bound'] == 'valu_issue':
        v = r['valu_issue']
        assert v['valu_busy'] >= 0.7 and 0.5 < v['frac'] <= 1.05 and v['peak_ginstr_per_s'] == pytest.approx(614.4)
    assert r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9
    # 8192 episodes, 20-step regions: ONE launch of the persistent one-wave kernel per region (regions of fewer than 10 steps, or CC4_PERSIST=0:
    # three or four concurrent launches of k_step_philox1 per step)
    assert r['step_kernel'] == 'k_step_philox1' and r['kernel'] == r['run_kernel'] == 'k_run_philox1' and r['steps_per_launch'] == 20
    assert 0 < r['step_ms'] <= d['ms_per_step'] * 1.02 and abs(r['launch_ms'] - 20 * r['step_ms']) < 1e-9
    assert abs(r['algorithmic_bytes_per_launch'] - 20 * r['algorithmic_bytes_per_step']) < 1e-6 * r['algorithmic_bytes_per_launch']
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['launch_ms'] * 1e-3) / 1e9) < 1e-6 * r['achieved']
    assert r['traffic'] is None or ('profiles/' in r['traffic_source'] and f"on {r['kernel']}:" in r['traffic_source'])     # only ever the timed kernel's own counters
    # r05: the consumable rates beside the closed loop -- actions written on the device every step, and the exchange on a one-rank communicator
    p = d['policy_in_loop']
    assert p['run_kernel'] == 'k_step_philox1' and 50e6 < p['value'] <= p['grouped']['value'] * 1.05 and p['grouped']['value'] < d['value'] * 1.05
    # r06: the rollout form beside it -- one launch of the persistent kernel's rollout build per region, the policy behind gates and publishes
    q = p['rollout']
    assert 'skipped' in q or (q['run_kernel'] == 'k_run_philox1r' and 50e6 < q['value'] < d['value'] * 1.05 and q['policy_ops_per_step'] >= 2)
    x = d['exchange_world1']
    assert 'skipped' in x or (x['envs_1024']['exchange']['in_kernel'] and x['envs_1024']['run_kernel'] == 'k_run_philox' and
                              x['envs_8192']['run_kernel'] == 'k_run_philox1x' and x['envs_8192']['exchange']['watchdog_timeouts'] == 0 and
                              x['envs_8192']['allgathers_issued'] >= x['envs_8192']['regions'] * 20)
    assert d['envs_1024']['roofline']['bound'] == 'latency' and d['single_env_facade']['us_per_step'] < 120
    assert abs(d['value'] - 5.0 * 8192 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value']
    pr = d['config']['per_rank']
    assert len(pr) == 1 and pr[0]['rank'] == 0 and pr[0]['envs'] == 8192 and pr[0]['host_launch_us_per_step'] > 0
    assert d['single_env_facade']['env_steps_per_sec'] > 34 and d['eval_sequential']['env_steps_per_sec'] > 34      # the reference's own rate
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and 'sample' in c and c['one_core']['cores'] == 1
    assert c['reference_python']['kind'] == 'reference' and c['reference_python']['value'] == 172.0
    assert d['alt_rng']['rng'] == 'pcg64' and d['envs_1024']['total_envs'] == 1024 and d['value'] > 50e6


def test_bench_line_of_long_regions():
    """`python bench.py --steps 64`: a timed region of 10 steps or more at 8192 episodes is ONE launch of the persistent kernel; the roofline
    object says so (kernel, steps_per_launch, launch_ms = the launch, step_ms = launch_ms / steps_per_launch)."""
    import json, os, subprocess, sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '64', '--warmup', '5', '--min-seconds', '0.05', '--no-alt', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.strip()][0])
    r = d['roofline']
    assert not d['config']['engine_error_flags'] and d['steps'] == 64
    assert r['step_kernel'] == 'k_step_philox1' and r['kernel'] == r['run_kernel'] == 'k_run_philox1' and r['steps_per_launch'] == 64
    assert 0 < r['step_ms'] <= d['ms_per_step'] * 1.02 and abs(r['launch_ms'] - 64 * r['step_ms']) < 1e-9
    assert abs(r['algorithmic_bytes_per_launch'] - 64 * r['algorithmic_bytes_per_step']) < 1e-6 * r['algorithmic_bytes_per_launch']
    assert abs(r['achieved'] - r['algorithmic_bytes_per_launch'] / (r['launch_ms'] * 1e-3) / 1e9) < 1e-6 * r['achieved']
    assert abs(d['value'] - 5.0 * 8192 / (d['ms_per_step'] * 1e-3)) < 1e-6 * d['value'] and d['value'] > 50e6
