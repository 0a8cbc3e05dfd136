"""CPU: host-side logic that needs no device -- sharding, observation/mask splitting, naming, spaces, argument checks."""
import numpy as np
import pytest
from cage_challenge_4_amd import shard_range, split_obs, split_mask
from cage_challenge_4_amd import wrappers as W
from cage_challenge_4_amd.spaces import Discrete, MultiDiscrete, MultiBinary
from oracle_binding import random_actions


def test_shard_range_partitions_exactly():
    for n in (1, 7, 8, 1000, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_split_layouts():
    obs = np.arange(3 * 578).reshape(3, 578)
    parts = split_obs(obs)
    assert [p.shape[1] for p in parts] == [92, 92, 92, 92, 210]
    assert parts[4][0, 0] == 368
    m = split_mask(np.zeros((2, 570), bool))
    assert [p.shape[1] for p in m] == [82, 82, 82, 82, 242]


def test_host_names_follow_reference_scheme():
    assert W.host_name(136) == 'root_internet_host_0'
    assert W.host_name(0) == 'restricted_zone_a_subnet_router'
    assert W.host_name(1) == 'restricted_zone_a_subnet_user_host_0'
    assert W.host_name(11) == 'restricted_zone_a_subnet_server_host_0'
    assert W.host_name(4 * 17 + 16) == 'contractor_network_subnet_server_host_5'


def test_scenario_generator_signature_and_constants():
    sg = W.EnterpriseScenarioGenerator(blue_agent_class=W.SleepAgent, green_agent_class=W.EnterpriseGreenAgent,
                                       red_agent_class=W.FiniteStateRedAgent, steps=500)
    assert (sg.MIN_USER_HOSTS, sg.MAX_USER_HOSTS, sg.MIN_SERVER_HOSTS, sg.MAX_SERVER_HOSTS) == (3, 10, 1, 6)
    assert sg.MESSAGE_LENGTH == 8 and sg.steps == 500
    sg2 = W.EnterpriseScenarioGenerator(blue_agent_class=W.SleepAgent, green_agent_class=W.SleepAgent,
                                        red_agent_class=W.DiscoveryFSRed)
    assert (sg2.red_policy, sg2.green_policy) == (2, 1)

    class NotAnAgent:                # no get_action: nothing the step could ask
        pass
    with pytest.raises(TypeError):
        W.EnterpriseScenarioGenerator(blue_agent_class=W.SleepAgent, green_agent_class=W.EnterpriseGreenAgent,
                                      red_agent_class=NotAnAgent)

    class KeyboardAgent:             # not one of the engine's built-in policies: its objects act from the host (r04), the device's red policy sleeps
        def get_action(self, observation, action_space):
            return None
    sg3 = W.EnterpriseScenarioGenerator(blue_agent_class=W.SleepAgent, green_agent_class=W.EnterpriseGreenAgent, red_agent_class=KeyboardAgent)
    assert sg3.custom == {'red': KeyboardAgent} and sg3.red_policy == 1


def test_spaces():
    assert Discrete(82).contains(81) and not Discrete(82).contains(82)
    md = MultiDiscrete([3] + [2] * 91)
    assert len(md) == 92 and md.contains(np.zeros(92, int))
    assert MultiBinary(8).contains(np.ones(8, bool)) and not MultiBinary(8).contains(np.ones(7, bool))


def test_random_action_generator_ranges_and_determinism():
    a = random_actions(1000, 3, 4096)
    assert a.shape == (4096, 5) and a.min() >= 0
    assert a[:, :4].max() <= 81 and a[:, 4].max() <= 241 and a[:, 4].max() > 200
    assert np.array_equal(a, random_actions(1000, 3, 4096))
    assert not np.array_equal(a, random_actions(1000, 4, 4096))
    # shard-invariance: env e draws the same action whatever batch it sits in
    assert np.array_equal(random_actions(1000 + 100, 3, 10), a[100:110])


def test_flat_obs_enumerations_agree():
    """env_flat_obs (12 parts, the oracle's encoder), env_flat_obs_at (by position) and env_flat_obs_sorted (by kind, what
    the device encodes) describe the same vector -- checked through the oracle build on a few evolved states."""
    import ctypes
    import numpy as np
    from oracle_binding import OracleVecEnv, random_actions
    o = OracleVecEnv(4, steps=60)
    o.reset(seeds=5)
    o.lib.cc4o_obs_variants.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    for t in range(40):
        obs, _, _, _ = o.step(random_actions(5, t, 4), np.random.default_rng(t).integers(0, 2, size=(4, 5, 8)).astype(np.uint8))
        for i in range(4):
            a = np.full(578, -1, np.int32)
            b = np.full(578, -1, np.int32)
            o.lib.cc4o_obs_variants(o._h, i, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p))
            assert np.array_equal(a, obs[i]) and np.array_equal(b, obs[i]), (t, i)
            c = np.full(578, -7, np.int32)                       # the table form of the values that can change every step
            o.lib.cc4o_obs_by_table.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
            o.lib.cc4o_obs_by_table(o._h, i, c.ctypes.data_as(ctypes.c_void_p))
            fast = c >= 0
            assert fast.sum() == 384 and np.array_equal(c[fast], obs[i][fast]), (t, i)


def test_builtin_policies_are_selected_by_class_name_and_foreign_classes_can_opt_out():
    """the reference's own CybORG.Agents.FiniteStateRedAgent (a class of another module with a built-in's name) must select
    the device policy, not silently fall to the host slow path; a class of any OTHER package that merely carries a built-in's name (a user's
    modified FiniteStateRedAgent) keeps its own get_action -- host path -- unless it opts into the device policy (cc4_device_policy = True);
    host_agents=True (or the class attribute cc4_host_agent) sends any class to the host on purpose; a class of another name always acts
    from the host."""
    import warnings

    class FiniteStateRedAgent:                     # a user's class that happens to carry a built-in's name
        def get_action(self, observation, action_space):
            return None
    FiniteStateRedAgent.__module__ = 'my_agents'
    sg = W.EnterpriseScenarioGenerator(red_agent_class=FiniteStateRedAgent)
    assert sg.custom == {'red': FiniteStateRedAgent} and sg.red_policy == 1       # its own get_action runs
    FiniteStateRedAgent.cc4_device_policy = True
    sg = W.EnterpriseScenarioGenerator(red_agent_class=FiniteStateRedAgent)
    assert sg.custom == {} and sg.red_policy == 0                                 # opted in: the device policy of that name
    del FiniteStateRedAgent.cc4_device_policy
    lookalike = type('FiniteStateRedAgent', (), {'get_action': lambda self, o, a: None})
    lookalike.__module__ = 'CybORGish.agents'                                     # (a prefix match on the package name is not enough)
    assert W.EnterpriseScenarioGenerator(red_agent_class=lookalike).custom == {'red': lookalike}
    ref_like = type('FiniteStateRedAgent', (), {'get_action': lambda self, o, a: None})
    ref_like.__module__ = 'CybORG.Agents.SimpleAgents.FiniteStateRedAgent'      # the reference's own class: no warning, the device policy
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        sg = W.EnterpriseScenarioGenerator(red_agent_class=ref_like)
    assert sg.custom == {} and sg.red_policy == 0 and not w
    sg = W.EnterpriseScenarioGenerator(red_agent_class=FiniteStateRedAgent, host_agents=True)
    assert sg.custom == {'red': FiniteStateRedAgent} and sg.red_policy == 1
    FiniteStateRedAgent.cc4_host_agent = True
    assert W.EnterpriseScenarioGenerator(red_agent_class=FiniteStateRedAgent).custom == {'red': FiniteStateRedAgent}
    other = type('ScriptedRed', (), {'get_action': lambda self, o, a: None})
    assert W.EnterpriseScenarioGenerator(red_agent_class=other).custom == {'red': other}
    with pytest.raises(TypeError):
        W.EnterpriseScenarioGenerator(red_agent_class=type('NoPolicy', (), {}))


def test_control_plane_never_hands_back_a_plane_of_another_kind(monkeypatch):
    """control_plane() caches one plane per process; a later call that asks for another kind (a SoloPlane cached at world
    size 1, then force=True) must raise instead of returning the cached one silently."""
    from cage_challenge_4_amd import distributed as D
    monkeypatch.setattr(D, '_PLANE', None)
    for k in ('RANK', 'WORLD_SIZE', 'CC4_CONTROL_PLANE'):
        monkeypatch.delenv(k, raising=False)
    solo = D.control_plane()
    assert isinstance(solo, D.SoloPlane) and D.control_plane() is solo
    with pytest.raises(RuntimeError, match="already runs a 'solo' plane"):
        D.control_plane(force=True)
    monkeypatch.setattr(D, '_PLANE', None)


def test_control_plane_repeated_calls_return_the_cached_file_or_gloo_plane(monkeypatch, tmp_path):
    """FilePlane and GlooPlane subclass SoloPlane, so classifying the cached plane with isinstance(.., SoloPlane) called every
    plane 'solo' and a second control_plane(force=True) -- or init_rccl's argument-less control_plane() at world > 1 -- raised.  file -> file,
    file -> default and gloo -> default return the cached object; file -> gloo still raises."""
    from cage_challenge_4_amd import distributed as D
    monkeypatch.setattr(D, '_PLANE', None)
    for k in ('RANK', 'WORLD_SIZE', 'CC4_CONTROL_PLANE'):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv('CC4_CONTROL_PLANE_KEY', f'hostlogic_{tmp_path.name}')
    f = D.control_plane(force=True)
    try:
        assert type(f) is D.FilePlane
        assert D.control_plane(force=True) is f            # file -> file
        assert D.control_plane() is f                      # file -> default: whatever exists
        assert D.control_plane(kind='file', force=True) is f
        with pytest.raises(RuntimeError, match="already runs a 'file' plane, a 'gloo' plane"):
            D.control_plane(kind='gloo', force=True)
        with pytest.raises(RuntimeError, match="already runs a 'file' plane, a 'solo' plane"):
            D.control_plane(kind='file')                   # names a kind at world 1 without force: that is a solo plane
    finally:
        f.close()
    g = object.__new__(D.GlooPlane)                        # (no process group: only the classification is under test)
    monkeypatch.setattr(D, '_PLANE', g)
    assert D.control_plane() is g                          # gloo -> default
    with pytest.raises(RuntimeError, match="already runs a 'gloo' plane, a 'file' plane"):
        D.control_plane(force=True)
    monkeypatch.setattr(D, '_PLANE', None)


def test_event_log_stays_on_for_backends_without_the_replay(oracle_lib):
    """The fixed-action wrappers switch the event log to on-demand only where the engine can repeat a step with the log on
    (cc4_keep_previous / cc4_replay_logged); on the CPU oracle of the tests it simply stays on, and dict observations keep working."""
    from oracle_binding import OracleVecEnv
    sg = W.EnterpriseScenarioGenerator(blue_agent_class=W.SleepAgent, green_agent_class=W.EnterpriseGreenAgent, red_agent_class=W.FiniteStateRedAgent, steps=30)
    env = W.BlueFlatWrapper(W.CybORG(sg, seed=5, vec_factory=OracleVecEnv))
    assert not env.env._lazy_log
    env.reset()
    env.step({a: 0 for a in env.possible_agents})
    assert 'success' in env.env.get_observation('blue_agent_0')


def test_cyborg_facade_methods_of_env_py(oracle_lib):
    """VERDICT r05 missing 3: CybORG.get_agent_ids (env.py:417), get_message_space (:449), get_observation_space (:284), start (:163),
    close (:426).  Expected values recorded from the reference in the build container: 60 ids at seed 5 in the order blue, green, red;
    get_observation_space raises NotImplementedError for an agent of the scenario and ValueError for any other name; start(n) steps n
    times without submitted actions and returns the done flag (False, then True once the episode ends)."""
    from oracle_binding import OracleVecEnv
    sg = W.EnterpriseScenarioGenerator(blue_agent_class=W.SleepAgent, green_agent_class=W.EnterpriseGreenAgent, red_agent_class=W.FiniteStateRedAgent, steps=20)
    env = W.CybORG(sg, seed=5, vec_factory=OracleVecEnv)
    ids = env.get_agent_ids()
    assert len(ids) == 60 and ids[:6] == [f'blue_agent_{b}' for b in range(5)] + ['green_agent_0']
    assert ids[-7:] == ['green_agent_48'] + [f'red_agent_{r}' for r in range(6)]
    assert set(env.active_agents) <= set(ids)
    sp = env.get_message_space('blue_agent_0')
    assert sp.n == 8 and sp.contains(np.zeros(8, dtype=np.int8))
    with pytest.raises(NotImplementedError):
        env.get_observation_space('blue_agent_0')
    with pytest.raises(ValueError, match='Agent nobody not in agent list'):
        env.get_observation_space('nobody')
    twin = W.CybORG(sg, seed=5, vec_factory=OracleVecEnv)
    assert env.start(3) is False
    for _ in range(3):
        twin.parallel_step({})
    assert env.get_rewards() == twin.get_rewards() and env.get_true_state() == twin.get_true_state()
    assert env.start(25) is True                     # the episode (20 steps) ends inside the call; the loop stops there
    env.close()
    env.close()
    twin.close()


def test_blue_slot_shape_is_the_wrappers_action_list(oracle_lib):
    """blue_slot_shape(b, idx) -- the one function the host decodes with and the device's slot table is filled from -- against the list layout of
    BlueFixedActionWrapper.py:241-300 written out here: Analyse x hosts, Monitor, Remove x hosts, Restore x hosts, Sleep, Allow x 8 per own subnet,
    Block x same, DeployDecoy x hosts; hosts = 6 server then 10 user positions of each own subnet in sorted order."""
    import ctypes
    oracle_lib.cc4o_blue_slot_shape.restype = ctypes.c_uint32
    oracle_lib.cc4o_blue_slot_shape.argtypes = [ctypes.c_int, ctypes.c_int]
    SLEEP, MONITOR, ANALYSE, REMOVE, RESTORE, DECOY, BLOCK, ALLOW = range(8)
    # subnet ids (cc4_state.h): restricted a 0, operational a 1, restricted b 2, operational b 3, contractor 4, public access 5, admin 6, office 7, internet 8
    sorted_subnets = [6, 4, 8, 7, 1, 3, 5, 0, 2]          # alphabetical: admin, contractor, internet, office, operational a/b, public, restricted a/b
    for b in range(5):
        own = [b] if b < 4 else [6, 7, 5]                 # agent 4: admin, office, public access -- sorted
        hosts = [17 * sn + (11 + k if k < 6 else 1 + k - 6) for sn in own for k in range(16)]
        pairs = [(dst, src) for dst in own for src in sorted_subnets if src != dst]
        want = ([(ANALYSE, h) for h in hosts] + [(MONITOR,)] + [(REMOVE, h) for h in hosts] + [(RESTORE, h) for h in hosts] + [(SLEEP,)]
                + [(ALLOW, d, s) for d, s in pairs] + [(BLOCK, d, s) for d, s in pairs] + [(DECOY, h) for h in hosts])
        assert len(want) == (82 if b < 4 else 242)
        for idx, w in enumerate(want):
            sh = oracle_lib.cc4o_blue_slot_shape(b, idx)
            got = (sh & 0xFF,) if w[0] in (SLEEP, MONITOR) else ((sh & 0xFF, (sh >> 8) & 0xFF) if len(w) == 2 else (sh & 0xFF, (sh >> 8) & 0xFF, sh >> 16))
            assert got == w, (b, idx, got, w)
        assert oracle_lib.cc4o_blue_slot_shape(b, len(want)) == SLEEP and oracle_lib.cc4o_blue_slot_shape(b, -1) == SLEEP


def test_monitor_roll_four_hosts_per_word_equals_the_per_host_function(oracle_lib):
    """monitor_roll4 (the lane-parallel kernels: one 32-bit word = four hosts' event bytes) against monitor_roll host by host, every byte value."""
    import ctypes
    oracle_lib.cc4o_monitor_roll4.restype = ctypes.c_uint32
    oracle_lib.cc4o_monitor_roll4.argtypes = [ctypes.c_uint32, ctypes.c_int]
    oracle_lib.cc4o_monitor_roll.restype = ctypes.c_uint32
    oracle_lib.cc4o_monitor_roll.argtypes = [ctypes.c_int, ctypes.c_uint32]
    rs = np.random.default_rng(5)
    for w in range(35):
        for ev4 in [0, 0xFFFFFFFF, 0x0F0F0F0F, 0x01020408] + [int(x) for x in rs.integers(0, 2**32, 40)]:
            ev4 &= 0x0F0F0F0F                       # the event bytes hold four bits
            got = oracle_lib.cc4o_monitor_roll4(ev4, w)
            for i in range(4):
                h = 4 * w + i
                b = (ev4 >> (8 * i)) & 0xFF
                want = oracle_lib.cc4o_monitor_roll(h, b) if h < 137 else b
                assert (got >> (8 * i)) & 0xFF == want, (w, i, hex(ev4))
