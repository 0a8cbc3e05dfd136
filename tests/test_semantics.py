"""CPU: behaviours the reference's own tests pin (CybORG/Tests/test_cc4/*), re-expressed against the oracle.
Each test names the reference test it follows."""
import os
import numpy as np
import pytest
from oracle_binding import OracleVecEnv

SORTED_SUBNETS = ['admin', 'contractor', 'internet', 'office', 'operational_a', 'operational_b', 'public_access',
                  'restricted_a', 'restricted_b']


def make(seed=123, steps=100, n=1):
    env = OracleVecEnv(n, steps=steps)
    env.reset(seeds=np.uint64(seed) + np.arange(n, dtype=np.uint64))
    return env


def sleep(n=1):
    return np.full((n, 5), -1, np.int32)


@pytest.mark.parametrize('steps,split', [(3, (1, 1, 1)), (10, (4, 3, 3)), (11, (4, 4, 3)), (100, (34, 33, 33))])
def test_mission_phase_split(steps, split):
    # test_mission_phase.py:95-190: phase index in obs[0] changes exactly at the split points; stepping past `steps` raises
    env = make(steps=steps)
    phases = []
    for t in range(steps):
        obs, *_ = env.step(sleep())
        phases.append(int(obs[0, 0]))
    want = [0] * split[0] + [1] * split[1] + [2] * split[2]
    assert phases == want
    *_, info = env.step(sleep())
    assert info['err'][0] & (1 << 7)       # reference: ValueError (State.py:539-540)


def test_done_turns_true_at_steps_minus_one():
    # EnterpriseScenarioGenerator.determine_done (ESG.py:862-870)
    env = make(steps=6)
    dones = [bool(env.step(sleep())[2][0]) for _ in range(6)]
    assert dones == [False, False, False, False, True, True]


def test_observation_layout_and_invariants():
    # test_BlueEnterpriseWrapper.py:30-45 slice layout
    env = make(seed=5, steps=50, n=8)
    rs = np.random.default_rng(0)
    for t in range(50):
        a = np.stack([rs.integers(0, 82, 8), rs.integers(0, 82, 8), rs.integers(0, 82, 8), rs.integers(0, 82, 8), rs.integers(0, 242, 8)], 1)
        obs, rew, done, info = env.step(a.astype(np.int32))
        assert info['err'].max() == 0
        for b, off, ns in ((0, 0, 1), (1, 92, 1), (2, 184, 1), (3, 276, 1), (4, 368, 3)):
            v = obs[:, off:off + (92 if ns == 1 else 210)]
            assert ((v[:, 0] >= 0) & (v[:, 0] <= 2)).all()
            assert ((v[:, 1:] == 0) | (v[:, 1:] == 1)).all()
            for k in range(ns):
                blk = v[:, 1 + 59 * k: 1 + 59 * (k + 1)]
                assert (blk[:, 0:9].sum(1) == 1).all()                       # own-subnet one-hot
                own = blk[:, 0:9].argmax(1)
                assert (blk[np.arange(8), 18 + own] == 1).all()             # comms policy: self bit is 1 (App. D)
                assert (blk[np.arange(8), 9 + own] == 0).all()              # a subnet never blocks itself
        assert (rew <= 0).all()


def test_own_subnet_positions_match_sorted_names():
    env = make()
    obs = env.reset(seeds=np.array([9], np.uint64))
    # agents 0..3: restricted_a, operational_a, restricted_b, operational_b ; agent 4: admin, office, public_access
    want = [[7], [4], [8], [5], [0, 3, 6]]
    for b, off in enumerate((0, 92, 184, 276, 368)):
        for k, pos in enumerate(want[b]):
            blk = obs[0, off + 1 + 59 * k: off + 1 + 59 * (k + 1)]
            assert blk[0:9].argmax() == pos


def test_block_and_allow_traffic_zone_bits():
    # test_BlueEnterpriseWrapper.py:232-254: after BlockTrafficZone the blocked bit of the source subnet is set
    env = make(seed=11)
    a = sleep()
    a[0, 0] = 58 + 2        # agent 0 (restricted_zone_a): Block <- third other subnet in sorted order = internet
    obs, *_ = env.step(a)
    blk = obs[0, 1:60]
    assert blk[9:18].tolist() == [0, 0, 1, 0, 0, 0, 0, 0, 0]
    a[0, 0] = 50 + 2        # Allow <- internet
    obs, *_ = env.step(a)
    assert obs[0, 1 + 9: 1 + 18].sum() == 0


def test_restore_costs_minus_one_on_submission_even_when_busy():
    # SURVEY Appendix B.3 (SimulationController.py:310), test_priority.py / README reward table
    env = make(seed=2)
    mask = env.mask()[0]
    idx = 33 + int(np.argmax(mask[33:49]))          # first valid Restore slot of agent 0
    base = OracleVecEnv(1, steps=100); base.reset(seeds=np.array([2], np.uint64))
    a = sleep(); a[0, 0] = idx
    for t in range(3):                              # Restore lasts 5 ticks: agent is busy on ticks 2,3
        r1 = env.step(a)[1][0]
        r0 = base.step(sleep())[1][0]
        assert r1 <= r0 - 1 + 1e-6 or t > 0        # first tick exactly -1 apart (same stream so far)
        if t == 0:
            assert r1 == r0 - 1.0


def test_invalid_host_slot_is_sleep_and_free():
    # BlueFixedActionWrapper.py:295-298: invalid slots map to Sleep (cost 0, no RNG)
    env = make(seed=4); ref = make(seed=4)
    mask = env.mask()[0]
    invalid = [i for i in range(33, 49) if not mask[i]]
    if not invalid:
        pytest.skip('zone is full at this seed')
    a = sleep(); a[0, 0] = invalid[0]
    for _ in range(5):
        o1, r1, *_ = env.step(a)
        o0, r0, *_ = ref.step(sleep())
        assert np.array_equal(o1, o0) and r1[0] == r0[0]
    assert np.array_equal(env.rng_state(), ref.rng_state())


def test_same_seed_same_trajectory_different_seed_differs():
    # test_cc4_seed.py:16-31 (stronger: whole observation stream)
    a, b, c = make(seed=77, steps=60), make(seed=77, steps=60), make(seed=78, steps=60)
    same, diff = True, False
    for _ in range(60):
        oa, ra, *_ = a.step(sleep()); ob, rb, *_ = b.step(sleep()); oc, rc, *_ = c.step(sleep())
        same &= np.array_equal(oa, ob) and ra[0] == rb[0]
        diff |= (not np.array_equal(oa, oc)) or ra[0] != rc[0]
    assert same and diff


def test_host_and_server_counts_within_readme_bounds():
    # test_Acceptance: 3..10 user hosts, 1..6 servers per subnet -> mask counts per zone
    env = make(seed=1, n=16)
    m = env.mask()
    for e in range(16):
        for b, off in enumerate((0, 82, 164, 246)):
            hosts = m[e, off:off + 16]                       # Analyse block: servers 0..5 then users 0..9
            assert 1 <= hosts[:6].sum() <= 6 and 3 <= hosts[6:].sum() <= 10
            assert hosts[0] and hosts[6]                     # server_host_0 / user_host_0 always exist
            assert m[e, off + 16] and m[e, off + 49]         # Monitor, Sleep
            assert m[e, off + 50:off + 66].all()             # Allow/Block always valid


def test_soak_regression_seeds_run_clean():
    # test_heuristic_agents.py:10-84 crash-regression seeds, random blue incl. invalid slots, 500 steps
    seeds = [6065, 5712, 9283, 3669, 4095, 87, 1148, 2742, 9556, 2812, 9251]
    env = OracleVecEnv(len(seeds), steps=500)
    env.reset(seeds=np.array(seeds, np.uint64))
    rs = np.random.default_rng(1)
    n = len(seeds)
    for t in range(500):
        a = np.stack([rs.integers(0, 82, n), rs.integers(0, 82, n), rs.integers(0, 82, n), rs.integers(0, 82, n), rs.integers(0, 242, n)], 1)
        obs, rew, done, info = env.step(a.astype(np.int32))
        assert info['err'].max() == 0, (t, info['err'])
    assert done.all()


def test_philox_and_pcg_modes_agree_in_distribution(oracle_lib):
    """BASELINE.md parity gate: the counter-based (Philox) mode draws at the same sites as the numpy-PCG64 mode, so episode
    statistics must agree.  384 independent 150-step SleepAgent-blue episodes per mode; compare mean episode reward and
    mean per-step fraction of hosts raising events with a z-test (|z| < 4; seeds are fixed, so this is deterministic)."""
    import ctypes
    n, T = 384, 150
    stats = {}
    for mode in (0, 1):
        env = OracleVecEnv(n, steps=T, rng_mode=mode)
        env.reset(seeds=np.uint64(50_000) + np.arange(n, dtype=np.uint64))
        acts = np.full((n, 5), -1, np.int32)
        ep = np.zeros(n); ev = np.zeros(n)
        for t in range(T):
            env.lib.cc4o_step_all(env._h, acts.ctypes.data_as(ctypes.c_void_p))
            for i in range(n):
                ep[i] += env.lib.cc4o_reward(env._h, i)
            if t % 10 == 9:
                for i in range(0, n, 8):
                    env._collect(i)
                    o = env._obs[i]
                    ev[i] += o[28:60].sum() + o[120:152].sum() + o[212:244].sum() + o[304:336].sum()
        assert max(env.lib.cc4o_err(env._h, i) for i in range(n)) == 0
        stats[mode] = (ep.mean(), ep.std(ddof=1) / np.sqrt(n), ev[::8].mean(), ev[::8].std(ddof=1) / np.sqrt(n / 8))
        env.close()
    z_rew = (stats[0][0] - stats[1][0]) / np.hypot(stats[0][1], stats[1][1])
    z_ev = (stats[0][2] - stats[1][2]) / max(np.hypot(stats[0][3], stats[1][3]), 1e-9)
    assert abs(z_rew) < 4.0, (stats, z_rew)
    assert abs(z_ev) < 4.0, (stats, z_ev)
    assert stats[0][0] < -100 and stats[1][0] < -100       # both modes actually play the game


def test_shared_topology_seed_semantics():
    """topology_seed (Philox mode): identical scenario across the batch at every (auto)reset, per-episode dynamics."""
    import numpy as np
    from oracle_binding import OracleVecEnv, random_actions
    o = OracleVecEnv(6, steps=25, rng_mode=1, autoreset=True, topology_seed=77)
    o.reset(seeds=300)
    first = o.topology(0).tobytes()
    assert all(o.topology(i).tobytes() == first for i in range(6))
    assert all(np.array_equal(o.get_state(0)[64:], o.get_state(i)[64:]) for i in range(1, 6))   # all but the generator key
    rew = []
    for t in range(30):
        rew.append(o.step(random_actions(300, t, 6))[1].copy())
    assert (np.array(rew).std(axis=1) > 0).any()
    again = o.topology(0).tobytes()
    assert again != first and all(o.topology(i).tobytes() == again for i in range(6))
    p = OracleVecEnv(6, steps=25, rng_mode=1, topology_seed=0)
    p.reset(seeds=300)
    assert len({p.topology(i).tobytes() for i in range(6)}) > 1                                     # default: per-episode topology


# comms policies of the three mission phases as the reference's own regression test states them
# (CybORG/Tests/test_cc4/test_issue22_blocks.py:24-55: rows/cols = [HQ(public, admin, office), contractor, RZA, OZA, RZB, OZB];
# 1 = green expects to communicate).  Data only.
_GROUPS = [[5, 6, 7], [4], [0], [1], [2], [3]]       # subnet indices (SUBNET enum order used by the engine)
_POLICY = [
    [[1, 1, 1, 0, 1, 0], [1, 1, 1, 0, 1, 0], [1, 1, 1, 1, 1, 0], [0, 0, 1, 1, 0, 0], [1, 1, 1, 0, 1, 1], [0, 0, 0, 0, 1, 1]],
    [[1, 1, 1, 0, 1, 0], [1, 1, 0, 0, 1, 0], [1, 0, 1, 0, 0, 0], [0, 0, 0, 1, 0, 0], [1, 1, 0, 0, 1, 1], [0, 0, 0, 0, 1, 1]],
    [[1, 1, 1, 0, 1, 0], [1, 1, 1, 0, 0, 0], [1, 1, 1, 1, 0, 0], [0, 0, 1, 1, 0, 0], [1, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 1]],
]


@pytest.mark.parametrize('mission_phase', [0, 1, 2])
def test_blocks_outside_the_comms_policy_do_not_touch_green(mission_phase, oracle_lib):
    """The property of the reference's test_issue22_blocks.py: with red asleep, blocking every subnet pair the phase's
    comms policy does not expect traffic on costs nothing -- green never picks those destinations.  Blocking a pair the
    policy does expect must cost something (the check that the test would notice a wrong allowed-subnet table)."""
    import ctypes
    from oracle_binding import OracleVecEnv
    buf = ctypes.create_string_buffer(8192)
    oracle_lib.cc4o_layout(buf, 8192)
    off = {ln.split()[0]: int(ln.split()[1]) for ln in buf.value.decode().splitlines() if len(ln.split()) == 2}

    def run(block_expected):
        o = OracleVecEnv(4, steps=300, red_policy=1)           # red_agent_class=SleepAgent
        o.reset(seeds=3)
        n = o.lib.cc4o_state_bytes()
        for i in range(4):
            st = np.frombuffer((ctypes.c_uint8 * n).from_address(o.lib.cc4o_state_ptr(o._h, i)), np.uint8)
            blocks = st[off['blocks']:off['blocks'] + 18].view(np.uint16)      # blocks[to] bit from
            for r, rg in enumerate(_GROUPS):
                for c, cg in enumerate(_GROUPS):
                    if _POLICY[mission_phase][r][c] == (1 if block_expected else 0) and (not block_expected or r != c):
                        for to in rg:
                            for fr in cg:
                                if to != fr:
                                    blocks[to] |= np.uint16(1 << fr)
            st[off['step_count']:off['step_count'] + 4].view(np.int32)[0] = 100 * mission_phase
        tot = 0.0
        for _ in range(99):
            tot += float(o.step(np.full((4, 5), -1, np.int32))[1].sum())
        return tot
    assert run(False) == 0.0
    assert run(True) < 0.0


def test_monitor_sees_events_on_the_server_host_and_on_client_hosts(oracle_lib):
    """The property of the reference's test_issue26_monitor.py: a network-connection / process-creation event is reported
    by the end-of-turn Monitor whether it sits on the blue agent's VelociraptorServer host or on a client host of the zone."""
    import ctypes
    from oracle_binding import OracleVecEnv
    buf = ctypes.create_string_buffer(8192)
    oracle_lib.cc4o_layout(buf, 8192)
    off = {ln.split()[0]: int(ln.split()[1]) for ln in buf.value.decode().splitlines() if len(ln.split()) == 2}
    o = OracleVecEnv(1, steps=300, red_policy=1, green_policy=1)          # everybody asleep: nothing else raises events
    o.reset(seeds=11)
    n = o.lib.cc4o_state_bytes()
    st = np.frombuffer((ctypes.c_uint8 * n).from_address(o.lib.cc4o_state_ptr(o._h, 0)), np.uint8)
    topo = o.topology(0)
    exists = lambda h: bool(topo[27 + 2 * h])                              # noqa: E731
    blue_base, blue_size = off['blue'], off['sizeof.BlueAgent']
    for b in range(4):                                                      # agents 0..3 own subnet b
        parent = int(st[blue_base + b * blue_size + 34])                   # BlueAgent.parent_host (csrc/cc4_state.h)
        assert parent // 17 == b and exists(parent)
        other = next(h for h in range(b * 17 + 1, b * 17 + 17) if exists(h) and h != parent)
        for h, bit in ((parent, 1), (other, 2)):                           # EV_CUR_CONN on one, EV_CUR_PROC on the other
            st[off['hev'] + h] |= bit                                      # EnvState.hev: one byte of EV_* bits per host
        obs = o.step(np.full((1, 5), -1, np.int32))[0][0]
        blk = obs[b * 92 + 1: b * 92 + 60]

        def slot(h):
            s_ = h % 17
            return (s_ - 11) if s_ >= 11 else 6 + (s_ - 1)                  # servers first, then users (BlueFlatWrapper order)
        assert blk[43 + slot(parent)] == 1 and blk[27 + slot(other)] == 1   # connection flag / process flag
        assert blk[27:59].sum() == 2


def test_counter_mode_scenario_generation_properties():
    """The per-host-phase generation of the counter-based mode (csrc/cc4_engine.h env_reset_counter_mode) keeps what
    EnterpriseScenarioGenerator guarantees: 3..10 user hosts and 1..6 servers per subnet, distinct addresses inside a subnet,
    network-wide unique service pids (_generate_pid, ESG.py:564-578), session pids above every service pid of their host
    (Host.create_pid, Host.py:198-200), and the same service statistics as the numpy-stream mode."""
    import json
    from oracle_binding import OracleVecEnv
    n = 192
    stats = {}
    for mode in (1, 0):
        o = OracleVecEnv(n, steps=100, rng_mode=mode)
        o.reset(seeds=4242)
        nsvc, nhosts, contested = [], [], 0
        for i in range(n):
            d = json.loads(o.true_state_json(i))
            hosts = d['hosts']
            by_subnet = {}
            pids = []
            for hd in hosts:
                h = hd['h']
                by_subnet.setdefault(h // 17, []).append(hd)
                svc_pids = [s[3] for s in hd['svcs']]
                pids += svc_pids
                assert all(1000 <= p < 10000 for p in svc_pids)
                for sess in ('blue', 'green'):
                    if hd.get(sess):
                        assert not svc_pids or hd[sess] > max(svc_pids)
                if h % 17 != 0:
                    nsvc.append(len(svc_pids))
            assert len(pids) == len(set(pids)), 'service pids are unique across the whole network'
            for sn, hs in by_subnet.items():
                if sn == 8:
                    assert [x['h'] for x in hs] == [136]
                    continue
                slots = [x['h'] % 17 for x in hs]
                users = [s for s in slots if 1 <= s <= 10]
                servers = [s for s in slots if s >= 11]
                assert 0 in slots and 3 <= len(users) <= 10 and 1 <= len(servers) <= 6
                assert users == list(range(1, 1 + len(users))) and servers == list(range(11, 11 + len(servers)))
                ips = [x['ip'] for x in hs]
                assert len(set(ips)) == len(ips) and all(1 <= ip <= 254 for ip in ips)
            assert len(set(d['cidr'])) == 9
            nhosts.append(len(hosts))
        stats[mode] = (np.mean(nsvc), np.mean(nhosts))
    # SSHD + Binomial-like add-ons (0..3 uniformly) + OT service in the two operational zones: the two modes agree closely
    assert abs(stats[0][0] - stats[1][0]) < 0.06 and abs(stats[0][1] - stats[1][1]) < 2.5, stats


def test_a_suspicious_pid_never_names_a_blue_or_green_session_process(oracle_lib):
    """VERDICT r02 missing #5.  StopProcess.kill_process (StopProcess.py:36-58) ends whatever session owns the killed pid, blue
    and green ones included; the engine only models the red case and raises E_BLUE_GREEN_SESSION_KILLED otherwise.  That flag
    is unreachable: the blue / green session processes are created by State.__init__ (State.py:103-136, Host.add_session ->
    create_pid, Host.py:189-200) and never removed or re-created (they are no service processes, and Host.restore puts them
    back under their original pids, Host.py:373-429), while every other process of a host is created later with pid =
    max(all current pids) + 1..9, i.e. above them -- and a suspicious pid is always the pid of such a later process (a red
    shell: the only events that carry a pid, ExploitAction.py:264-275).  Checked here on what the engine itself produces under
    the blue policies that make pids go stale (Remove / Restore heavy): every entry of every sus list exceeds the blue and green
    session pids of its host, for whole episodes."""
    import json
    n, T = 48, 300
    ora = OracleVecEnv(n, steps=T); ora.reset(seeds=6100)
    rs = np.random.default_rng(3)
    mask = ora.mask()
    for t in range(T - 1):
        a = np.zeros((n, 5), np.int32)
        for b in range(5):
            nh, nc = (48, 24) if b == 4 else (16, 8)
            kind = rs.integers(0, 3, size=n)                      # Remove / Restore / DeployDecoy on a random valid host
            base = np.where(kind == 0, nh + 1, np.where(kind == 1, 2 * nh + 1, 3 * nh + 2 + 2 * nc))
            off = 82 * b if b < 4 else 328
            for e in range(n):
                valid = np.nonzero(mask[e, off + base[e]: off + base[e] + nh])[0]
                a[e, b] = base[e] + valid[rs.integers(len(valid))]
        ora.step(a)
        assert not ora._err.any(), t
        if t % 25 == 24 or t == T - 2:
            for e in range(0, n, 5):
                st = json.loads(ora.true_state_json(e))
                floor = {h['h']: max(h['blue'], h['green']) for h in st['hosts']}
                entries = [(h, p) for b in st['blue'] for h, p in b['sus']]
                assert all(p > floor[h] for h, p in entries), (t, e)
    assert sum(len(b['sus']) for b in json.loads(ora.true_state_json(0))['blue']) > 0


@pytest.mark.parametrize('rng_mode', [0, 1])
def test_every_host_row_a_step_changes_is_marked_dirty(oracle_lib, rng_mode):
    """k_step_philox (the four-wave kernel, the whole row in LDS) writes back the agent part and only those 64-byte host-table
    rows (HostDyn) the engine marked in StepWork.hdirty (hd_touch at each write site).  A row changed without a mark would be
    lost on the GPU only and only steps later -- so the marks are checked here on the CPU build of the same engine source:
    cc4o_step_check_marks compares every HostDyn row before and after each step.  Action mix: the blue actions that write the
    table (Remove / Restore / DeployDecoy) against the red FSM and green traffic, whole episodes; and the marks must be
    selective (a handful of the 137 rows per step), or the kernel would gain nothing from them."""
    n, T = 32, 260
    ora = OracleVecEnv(n, steps=T, rng_mode=rng_mode); ora.reset(seeds=7300)
    ora.check_marks = True
    rs = np.random.default_rng(5)
    mask = ora.mask()
    for t in range(T - 1):
        a = np.zeros((n, 5), np.int32)
        for b in range(5):
            nh, nc = (48, 24) if b == 4 else (16, 8)
            kind = rs.integers(0, 4, size=n)                      # Sleep-ish / Remove / Restore / DeployDecoy on a random valid host
            base = np.where(kind <= 1, nh + 1, np.where(kind == 2, 2 * nh + 1, 3 * nh + 2 + 2 * nc))
            off = 82 * b if b < 4 else 328
            for e in range(n):
                valid = np.nonzero(mask[e, off + base[e]: off + base[e] + nh])[0]
                a[e, b] = 0 if (kind[e] == 0 and t % 3) else base[e] + valid[rs.integers(len(valid))]
        ora.step(a)
    assert ora.unmarked_rows == 0, ora.unmarked_rows
    per_step = ora.marked_rows / (n * (T - 1))
    assert 0.5 < per_step < 12, per_step


def _engine_marginals(env_cls, n, seed0, chunk=2000, **kw):
    import gen_marginals as GM
    acc = GM.empty()
    done = 0
    while done < n:
        m = min(chunk, n - done)
        env = env_cls(m, steps=500, rng_mode=1, **kw)
        env.reset(seeds=np.uint64(seed0 + done) + np.arange(m, dtype=np.uint64))
        for i in range(m):
            GM.accumulate(acc, GM.desc_from_true_state(env.true_state_json(i)))
        env.close()
        done += m
    return acc


def check_generation_marginals(acc, p_floor=1e-4):
    """Gate: no duplicate service pid in any scenario (exact), and no marginal whose two-sample chi-square against the
    reference's 10 000 scenarios has p < p_floor (35 tests: a correct generator trips the gate with probability < 0.4 %).
    Power: the per-host marginals (service mix, OS, pid range, session pid offsets, addresses) rest on > 400 000 engine hosts
    against > 800 000 reference hosts, so a cell probability that is off by 0.5 % (absolute) is rejected with probability > 0.99;
    the per-scenario marginals (host counts per subnet, 8 and 6 cells; parent / start host kinds) rest on the scenario counts,
    and reject a cell that is off by 3 % (absolute) with probability > 0.95."""
    import json
    import os
    import gen_marginals as GM
    import golden_util as G
    ref = GM.from_json(json.load(open(os.path.join(G.GOLDEN_DIR, 'gen_marginals_ref.json'))))
    assert ref['scenarios'] >= 10000 and ref['duplicate_pid_scenarios'] == 0
    assert acc['duplicate_pid_scenarios'] == 0                                   # pids are unique network-wide, as _generate_pid makes them
    res = GM.compare(ref, acc)
    assert len(res) >= 35
    bad = {k: v for k, v in res.items() if v[2] < p_floor}
    assert not bad, bad
    return res


def test_counter_mode_generation_matches_the_reference_marginals(oracle_lib):
    """VERDICT r02 #2 (reset half): the counter mode generates scenarios per host and resolves pid collisions against all hosts
    at once -- not the reference's serial draw order, so there is no trajectory to replay; its output is gated on the marginals
    of 10 000 reference scenarios instead (tests/golden/gen_marginals_ref.json, oracle/refgen/make_gen_marginals.py): hosts per
    subnet, ordered add-on service mix (with / without the OT service), OS, pid range and uniqueness, session pid offsets,
    parent / start host kinds, red_agent_5's start subnet, subnet and host addresses."""
    acc = _engine_marginals(OracleVecEnv, 5000, 900000)
    res = check_generation_marginals(acc)
    assert acc['scenarios'] == 5000 and acc['hosts_total'] > 400000
    # ... and the numpy-stream generation (bit-exact with the reference on the golden seeds) passes the same gate on other seeds
    import gen_marginals as GM
    pcg = GM.empty()
    env = OracleVecEnv(1500, steps=500, rng_mode=0)
    env.reset(seeds=np.uint64(700000) + np.arange(1500, dtype=np.uint64))
    for i in range(1500):
        GM.accumulate(pcg, GM.desc_from_true_state(env.true_state_json(i)))
    check_generation_marginals(pcg)


def test_fsm_agent_reading_a_hostname_keyed_observation_of_an_unknown_host_is_on_the_references_crash_path(oracle_lib):
    """VERDICT r05 missing 4.  A FiniteStateRedAgent that reads the observation of a SUBMITTED hostname-keyed action (PrivilegeEscalate / Impact /
    DegradeServices through the `actions` dict) on a host whose hostname it has never seen: the reference files the host under host_states[None]
    (FiniteStateRedAgent.py:190-236) -- no exception at that step -- and is then on a crash path: the next DiscoverRemoteSystems result it processes
    raises ipaddress.AddressValueError (IPv4Address(None), :141-143), the phantom being chosen raises TypeError (:297-299 / :113).  Recorded from the
    reference's own class by oracle/refgen/make_fsm_phantom_golden.py.  The engine flags the step that puts the agent on that path (E_UNREACHABLE,
    csrc/cc4_engine.h fsm_observe): the wrappers raise CC4EngineError there -- one to a few steps before the reference's agent would have."""
    import json
    from conftest import ROOT
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'fsm_phantom_host.json')))
    assert g['at_the_observation'] == 'no exception' and g['host_states_keys'][-1] == 'None' and g['phantom']['state'] == 'K'
    assert g['next_discover_remote_systems_result'] == 'AddressValueError' and g['phantom_chosen'] == 'TypeError'
    src = open(os.path.join(ROOT, 'cage_challenge_4_amd', 'csrc', 'cc4_engine.h')).read()
    i = src.index("// ip looked up through a known hostname; unknown -> reference would key host_states[None]")
    assert 'set_err(x, E_UNREACHABLE)' in src[i:i + 300]
