"""CC4VecEnv -- N independent CC4 episodes stepped by the HIP engine (libcc4.so) on one MI355X.

Vectorised counterpart of  BlueFlatWrapper(CybORG(EnterpriseScenarioGenerator(SleepAgent, FiniteStateRedAgent,
EnterpriseGreenAgent, steps), seed))  (reference: CybORG/env.py:53-123, CybORG/Agents/Wrappers/BlueFlatWrapper.py).
Observations are returned as one int32 array [N, 578] = agents 0..3 (92 each) then agent 4 (210);
`split_obs` gives the per-agent views.  Rewards are the team reward every blue agent receives."""
import ctypes
import numpy as np
from . import _lib as L

RNG_PCG64 = 0    # numpy Generator(PCG64(SeedSequence(seed))): bit-exact with the reference under the same seed
RNG_PHILOX = 1   # Philox4x32-10 keyed by seed, counter (draw, stream, step, episode)
RED_FSM, RED_SLEEP, RED_DISCOVERY, RED_RANDOM = 0, 1, 2, 3   # red_agent_class: FiniteStateRedAgent / SleepAgent / DiscoveryFSRed / RandomSelectRedAgent
GREEN_ENTERPRISE, GREEN_SLEEP = 0, 1            # green_agent_class: EnterpriseGreenAgent / SleepAgent

ERR_NAMES = {0: 'PROC_OVERFLOW', 1: 'RSESS_OVERFLOW', 2: 'KNOWN_SID_OVERFLOW', 3: 'KNOWLEDGE_BLOCK_OVERFLOW',
             4: 'SUS_OVERFLOW', 5: 'OBS_OVERFLOW', 6: 'PENDING_EVENT_OVERFLOW', 7: 'STEP_PAST_END',
             8: 'UNREACHABLE_REFERENCE_PATH', 10: 'BLUE_GREEN_SESSION_KILLED', 11: 'FSM_NO_HOST'}


def pcg64_words(gen):
    """numpy Generator(PCG64) -> the six words cc4_set_rng_state takes."""
    st = gen.bit_generator.state
    if st.get('bit_generator') != 'PCG64':
        raise NotImplementedError(f"only numpy Generator(PCG64) streams can be adopted by the device RNG (got {st.get('bit_generator')})")
    m = (1 << 64) - 1
    return [st['state']['state'] >> 64, st['state']['state'] & m, st['state']['inc'] >> 64, st['state']['inc'] & m,
            int(st['has_uint32']), int(st['uinteger'])]


class CC4EngineError(RuntimeError):
    """An episode left the part of the reference's behaviour the engine reproduces (a fixed-size container overflowed, or the
    reference itself would have crashed): its results can no longer be trusted to equal the reference's."""


def err_names(bits):
    return [ERR_NAMES.get(i, f'bit{i}') for i in range(32) if (int(bits) >> i) & 1]


def raise_on_engine_error(err):
    """err: uint32 flags per episode (cc4_get_err).  Bit 7 is the reference's own ValueError (State.py:539-540); any other bit
    raises CC4EngineError -- nothing is ever dropped silently."""
    err = np.asarray(err)
    if not err.any():
        return
    if (err & (1 << 7)).any():
        raise ValueError("Step number exceeds last mission phase step maximum. "
                         "Use step parameter in EnterpriseScenarioGenerator.")  # State.py:539-540
    bad = np.nonzero(err)[0]
    raise CC4EngineError(f"engine error flags on {bad.size} episode(s), first: episode {int(bad[0])} {err_names(err[bad[0]])} "
                         f"(see ERR_NAMES / include/cc4.h; results of a flagged episode may differ from the reference)")


class CC4VecEnv:
    def __init__(self, num_envs, steps=500, rng_mode=RNG_PCG64, device_id=0, autoreset=False, red_policy=0, green_policy=0,
                 topology_seed=0, strict=True, blue_policy=0):
        """topology_seed != 0 (RNG_PHILOX only): all episodes share the scenario drawn from that key (uniform topology).
        blue_policy 1: cc4BlueRandomAgent acts for every blue agent whose action index is negative (0: such an agent sleeps).
        strict: raise (ValueError for a step past the episode's end, as the reference does; CC4EngineError otherwise) as soon as
        any episode carries an error flag.  strict=False leaves the flags to the caller (`err`, `info['err']`)."""
        self.strict = bool(strict)
        self.lib = L.load()
        self.num_envs = int(num_envs)
        self.steps = int(steps)
        cfg = L.CC4Config(self.num_envs, self.steps, int(device_id), int(rng_mode), int(bool(autoreset)),
                          int(red_policy), int(green_policy), int(topology_seed), int(blue_policy))
        h = ctypes.c_void_p()
        rc = self.lib.cc4_create(ctypes.byref(cfg), ctypes.byref(h))
        if rc != 0:
            msg = self.lib.cc4_last_error(h if h else None)
            if h:
                self.lib.cc4_destroy(h)
            raise L.CC4Error(f"cc4_create failed (rc={rc}): {msg.decode() if msg else ''}")
        self._h = h
        n = self.num_envs
        self._obs = np.zeros((n, L.OBS_PER_ENV), np.int32)
        self._rew = np.zeros(n, np.float32)
        self._done = np.zeros(n, np.uint8)
        self._mask = np.zeros((n, L.MASK_PER_ENV), np.uint8)
        self._err = np.zeros(n, np.uint32)
        self._err_seen = np.zeros(n, np.uint32)      # flags already raised for (strict mode raises once per flag and episode)
        self._any_seen = False
        vp = ctypes.c_void_p
        self._p_out = (self._obs.ctypes.data_as(vp), self._rew.ctypes.data_as(vp), self._done.ctypes.data_as(vp), self._err.ctypes.data_as(vp))

    # -- lifecycle
    def close(self):
        if getattr(self, '_h', None):
            self.lib.cc4_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        L.check(self.lib, self._h, rc, what)

    # -- API
    def reset(self, seeds=None, env_mask=None):
        """seeds: None (streams continue, == reference reset(seed=None)), int (seed+i per env) or array [N]."""
        sp = None
        if seeds is not None:
            if np.isscalar(seeds):
                seeds = np.uint64(seeds) + np.arange(self.num_envs, dtype=np.uint64)
            seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
            assert seeds.shape == (self.num_envs,)
            sp = seeds.ctypes.data_as(ctypes.c_void_p)
        mp = None
        if env_mask is not None:
            env_mask = np.ascontiguousarray(env_mask, dtype=np.uint8)
            mp = env_mask.ctypes.data_as(ctypes.c_void_p)
        self._chk(self.lib.cc4_reset(self._h, sp, mp), 'cc4_reset')
        return self._fetch(mask=True)[0]

    def step(self, actions=None, messages=None):
        """actions: int array [N,5] of wrapper indices (None / negative = no action -> Sleep)."""
        ap = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.int32)
            assert actions.shape == (self.num_envs, L.NUM_BLUE)
            ap = actions.ctypes.data_as(ctypes.c_void_p)
        mp = None
        if messages is not None:
            messages = np.ascontiguousarray(messages, dtype=np.uint8)
            assert messages.shape == (self.num_envs, L.NUM_BLUE, L.MSG_LEN)
            mp = messages.ctypes.data_as(ctypes.c_void_p)
        # cc4_step + cc4_fetch in one call: inputs up in one copy, the launch, the four results down in one copy, one host wait
        rc = self.lib.cc4_step_fetch(self._h, ap, mp, *self._p_out)
        if rc:
            self._chk(rc, 'cc4_step_fetch')
        obs, rew, done = self._check_err()
        return obs, rew, done, {'err': self._err}

    def agent_actions(self, kind):
        """An all-CC4_ACT_NONE array of cc4_agent_action records for step_ex: [N, 6] ('red') or [N, 80] ('green'); numpy
        structured dtype with the fields of include/cc4.h (type, host, arg, ticks, session, flags, rate0, rate1)."""
        a = np.zeros((self.num_envs, L.NUM_RED if kind == 'red' else L.MAX_GREEN), dtype=L.AGENT_ACTION_DTYPE)
        a['type'] = -1
        return a

    def step_ex(self, actions=None, messages=None, red=None, green=None):
        """cc4_step_ex: a step that also takes the red / green entries of the reference's `actions` dict (SimulationController.py:
        236-240) as agent_actions() arrays.  The blue indices may carry `action.duration` in bits 20..27."""
        ap = mp = rp = gp = None
        if actions is not None:
            actions = np.ascontiguousarray(actions, dtype=np.int32)
            assert actions.shape == (self.num_envs, L.NUM_BLUE)
            ap = actions.ctypes.data_as(ctypes.c_void_p)
        if messages is not None:
            messages = np.ascontiguousarray(messages, dtype=np.uint8)
            assert messages.shape == (self.num_envs, L.NUM_BLUE, L.MSG_LEN)
            mp = messages.ctypes.data_as(ctypes.c_void_p)
        if red is not None:
            red = np.ascontiguousarray(red, dtype=L.AGENT_ACTION_DTYPE)
            assert red.shape == (self.num_envs, L.NUM_RED) and red.dtype.itemsize == 24
            rp = red.ctypes.data_as(ctypes.c_void_p)
        if green is not None:
            green = np.ascontiguousarray(green, dtype=L.AGENT_ACTION_DTYPE)
            assert green.shape == (self.num_envs, L.MAX_GREEN) and green.dtype.itemsize == 24
            gp = green.ctypes.data_as(ctypes.c_void_p)
        self._chk(self.lib.cc4_step_ex(self._h, ap, mp, rp, gp), 'cc4_step_ex')
        obs, rew, done = self._fetch()
        return obs, rew, done, {'err': self._err}

    def edit_state(self, env, op, a0=0, a1=0, a2=0):
        """cc4_edit_state: a direct edit of one episode between steps (what the reference's scripted tests do to
        env.environment_controller.state by hand): op 0 mission phase, 1 add service, 2 service reliability, 3 clear host,
        4 deploy one decoy kind, 5 add a red session, 6 block / allow a subnet pair, 7 step count, 8 add a host event, 9 a red
        agent's `active` flag (include/cc4.h).  Returns the op's result (>= 0)."""
        rc = int(self.lib.cc4_edit_state(self._h, int(env), int(op), int(a0), int(a1), int(a2)))
        if rc < 0:
            self._chk(rc, 'cc4_edit_state')
        return rc

    def _fetch(self, mask=False):
        self._chk(self.lib.cc4_fetch(self._h, *self._p_out), 'cc4_fetch')      # observations, reward, done, error flags: one copy
        if mask:
            self._chk(self.lib.cc4_get_action_mask(self._h, self._mask.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_action_mask')
        return self._check_err()

    def _check_err(self):
        if not self._any_seen and not self._err.any():      # the common case: no flag anywhere, none remembered
            return self._obs, self._rew, self._done.astype(bool)
        self._err_seen &= self._err                    # a regenerated episode (reset, autoreset) starts with a clean slate
        if self.strict:
            # a flag stays set in the episode's row until the episode is reset; it is raised ONCE -- a batch of thousands of
            # episodes is not taken hostage by one of them: the caller may reset that episode (env_mask) or carry on, `err` and
            # info['err'] keep showing the flag.  (A step past the episode's end raises on every call, as the reference does.)
            fresh = self._err & ~self._err_seen
            self._err_seen |= self._err
            self._any_seen = bool(self._err_seen.any())
            raise_on_engine_error(fresh | (self._err & np.uint32(1 << 7)))
        elif (self._err & (1 << 7)).any():
            raise ValueError("Step number exceeds last mission phase step maximum. "
                             "Use step parameter in EnterpriseScenarioGenerator.")  # State.py:539-540
        return self._obs, self._rew, self._done.astype(bool)

    @property
    def action_mask(self):
        return self._mask.astype(bool)

    @property
    def err(self):
        return self._err

    def topology(self, env=0):
        """cc4_get_topology: 9 cidr octets, 9 user counts, 9 server counts, 137 x (exists, ip octet)."""
        buf = np.zeros(L.TOPOLOGY_BYTES, np.uint8)
        self._chk(self.lib.cc4_get_topology(self._h, int(env), buf.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_topology')
        return buf

    def enable_event_log(self, on=True):
        """cc4_enable_event_log: record the HostEvents entries of every step (for decoded blue dict observations)."""
        self._chk(self.lib.cc4_enable_event_log(self._h, int(bool(on))), 'cc4_enable_event_log')

    def keep_previous(self, on=True):
        """cc4_keep_previous: keep the rows as they stood before every step, so that replay_logged() can produce the last step's event log on demand."""
        self._chk(self.lib.cc4_keep_previous(self._h, int(bool(on))), 'cc4_keep_previous')

    def replay_logged(self):
        """cc4_replay_logged: the last step once more on the kept rows with the event log on; true_state_json then carries its events."""
        self._chk(self.lib.cc4_replay_logged(self._h), 'cc4_replay_logged')

    def true_state_json(self, env=0):
        """cc4_get_true_state: the episode's packed state as the JSON document described in csrc/cc4_export.h."""
        need = int(self.lib.cc4_get_true_state(self._h, int(env), None, 0))
        self._chk(0 if need > 0 else need, 'cc4_get_true_state')
        buf = ctypes.create_string_buffer(need)
        self._chk(0 if self.lib.cc4_get_true_state(self._h, int(env), buf, need) > 0 else -1, 'cc4_get_true_state')
        return buf.value.decode()

    def set_seed(self, seeds):
        """cc4_set_seed: fresh generators (CybORG.set_seed) without touching the episodes."""
        if np.isscalar(seeds):
            seeds = np.uint64(seeds) + np.arange(self.num_envs, dtype=np.uint64)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        assert seeds.shape == (self.num_envs,)
        self._chk(self.lib.cc4_set_seed(self._h, seeds.ctypes.data_as(ctypes.c_void_p)), 'cc4_set_seed')

    def set_generators(self, generators):
        """cc4_set_rng_state: adopt the streams of numpy Generator(PCG64) objects (one per episode).  The objects themselves are
        not advanced afterwards: the stream continues on the device."""
        w = np.zeros((self.num_envs, 6), np.uint64)
        for i, g in enumerate(generators):
            w[i] = pcg64_words(g)
        self._chk(self.lib.cc4_set_rng_state(self._h, w.ctypes.data_as(ctypes.c_void_p)), 'cc4_set_rng_state')

    @property
    def step_kernel(self):
        """cc4_step_kernel: the kernel this handle's steps launch ('k_step', 'k_step_philox' or 'k_step_philox1')."""
        return self.lib.cc4_step_kernel(self._h).decode()

    @property
    def run_kernel(self):
        """cc4_run_kernel: the kernel run_random_steps launches ('k_run_philox' = one launch for all k steps of a small batch)."""
        return self.lib.cc4_run_kernel(self._h).decode()

    def run_kernel_for(self, k):
        """cc4_run_kernel_for: the kernel a run_random_steps call of k steps launches."""
        return self.lib.cc4_run_kernel_for(self._h, int(k)).decode()

    @property
    def launches_per_step(self):
        """cc4_launches_per_step: a step of a large batch is several launches (episode groups on separate streams)."""
        return int(self.lib.cc4_launches_per_step(self._h))

    def host_stats(self):
        """cc4_host_stats as a dict."""
        out = np.zeros(5, np.float64)
        self._chk(self.lib.cc4_host_stats(self._h, out.ctypes.data_as(ctypes.c_void_p)), 'cc4_host_stats')
        return {'steps': int(out[0]), 'launch_us': float(out[1]), 'gather_us': float(out[2]), 'gathers': int(out[3]), 'gather_stalls': int(out[4])}

    def exchange_info(self):
        """cc4_exchange_info: the exchange inside the one-launch kernels (on / off, ring depth, steps per publish, calls served, watchdog firings)."""
        out = np.zeros(5, np.int32)
        self._chk(self.lib.cc4_exchange_info(self._h, out.ctypes.data_as(ctypes.c_void_p)), 'cc4_exchange_info')
        return {'in_kernel': bool(out[0]), 'ring': int(out[1]), 'chunk': int(out[2]), 'calls': int(out[3]), 'watchdog_timeouts': int(out[4])}

    def verify_stats(self):
        """cc4_verify_stats (CC4_PERSIST_VERIFY=1): (one-launch calls checked against per-step launches, calls that disagreed)."""
        out = (ctypes.c_int64 * 2)()
        self._chk(self.lib.cc4_verify_stats(self._h, out), 'cc4_verify_stats')
        return int(out[0]), int(out[1])

    def gather_log(self, steps):
        """cc4_debug_gather_log: keep the gathered rows of the next `steps` in-kernel-exchanged steps (0: free the log)."""
        self._chk(self.lib.cc4_debug_gather_log(self._h, int(steps)), 'cc4_debug_gather_log')

    def get_gather_log(self, world, first, count):
        """[count, world * N, 148] packed rows of steps first .. first + count - 1 of the log (distributed.unpack_obs decodes them)."""
        out = np.zeros((count, world * self.num_envs, 148), np.uint8)
        rc = self.lib.cc4_get_gather_log(self._h, out.ctypes.data_as(ctypes.c_void_p), int(first), int(count))
        if rc < 0:
            self._chk(rc, 'cc4_get_gather_log')
        return out

    def comm_info(self):
        """cc4_comm_info: RCCL's view of the communicator (ranks it spans, this rank, its device) + the identity of the handle's device."""
        out = np.zeros(8, np.int32)
        uuid = ctypes.create_string_buffer(33)
        self._chk(self.lib.cc4_comm_info(self._h, out.ctypes.data_as(ctypes.c_void_p), uuid), 'cc4_comm_info')
        return {'nccl_comm_count': int(out[0]), 'nccl_user_rank': int(out[1]), 'nccl_device': int(out[2]), 'hip_device': int(out[3]),
                'pci': f'{int(out[4]):04x}:{int(out[5]):02x}:{int(out[6]):02x}', 'compute_units': int(out[7]), 'device_uuid': uuid.value.decode()}

    def rng_state(self):
        out = np.zeros((self.num_envs, 7), np.uint64)
        self._chk(self.lib.cc4_get_rng_state(self._h, out.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_rng_state')
        return out

    def get_state(self, env):
        buf = np.zeros(self.lib.cc4_state_bytes(), np.uint8)
        self._chk(self.lib.cc4_get_state(self._h, int(env), buf.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_state')
        return buf

    def get_states(self):
        """The packed hot rows of ALL episodes, [num_envs][cc4_state_bytes] (checkpointing a batch, parity tests)."""
        nb = int(self.lib.cc4_state_bytes())
        out = np.zeros((self.num_envs, nb), np.uint8)
        self._chk(self.lib.cc4_get_states(self._h, 0, self.num_envs, out.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_states')
        return out

    def get_cold(self, env):
        cold = np.zeros(self.lib.cc4_cold_bytes(self._h), np.uint8)
        self._chk(self.lib.cc4_get_cold(self._h, int(env), cold.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_cold')
        return cold

    def set_state(self, env, buf):
        buf = np.ascontiguousarray(buf, dtype=np.uint8)
        assert buf.size == self.lib.cc4_state_bytes()
        self._chk(self.lib.cc4_set_state(self._h, int(env), buf.ctypes.data_as(ctypes.c_void_p)), 'cc4_set_state')

    def snapshot(self, env):
        """Full episode snapshot (hot + cold row) for checkpointing / tree search / parity bisecting."""
        cold = np.zeros(self.lib.cc4_cold_bytes(self._h), np.uint8)
        self._chk(self.lib.cc4_get_cold(self._h, int(env), cold.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_cold')
        return self.get_state(env), cold

    def restore(self, env, snap):
        hot, cold = snap
        self.set_state(env, hot)
        cold = np.ascontiguousarray(cold, dtype=np.uint8)
        assert cold.size == self.lib.cc4_cold_bytes(self._h)
        self._chk(self.lib.cc4_set_cold(self._h, int(env), cold.ctypes.data_as(ctypes.c_void_p)), 'cc4_set_cold')

    # device-resident loop used by bench.py
    def run_random_steps(self, seed0, t0, k, timed=True):
        ms = ctypes.c_float(0.0)
        self._chk(self.lib.cc4_run_random_steps(self._h, ctypes.c_uint64(seed0), ctypes.c_uint32(t0), int(k),
                                                ctypes.byref(ms) if timed else None), 'cc4_run_random_steps')
        return float(ms.value)

    def run_policy_steps(self, seed0, t0, k):
        """A learner's loop without the learner: per step a kernel writes the [N, 5] action indices into the handle's device buffer
        (cc4_random_actions_device -- k_random_actions standing in for a policy) and cc4_step_device consumes them; no host synchronisation,
        every step's observations stay readable on the device (cc4_obs_device).  Same draws as run_random_steps."""
        p = ctypes.c_void_p()
        self._chk(self.lib.cc4_actions_device(self._h, ctypes.byref(p)), 'cc4_actions_device')
        s0 = ctypes.c_uint64(seed0)
        for i in range(int(k)):
            rc = self.lib.cc4_random_actions_device(self._h, s0, ctypes.c_uint32(t0 + i)) or self.lib.cc4_step_device(self._h, p, None)
            if rc:
                self._chk(rc, 'cc4_step_device')
        return 0.0

    def run_policy_steps_grouped(self, seed0, t0, k):
        """The same loop with the policy applied per episode group on the group's own stream (cc4_group_info /
        cc4_random_actions_group_device / cc4_step_group_device): no cross-stream dependency, the groups' pipelines stay apart."""
        p = ctypes.c_void_p()
        self._chk(self.lib.cc4_actions_device(self._h, ctypes.byref(p)), 'cc4_actions_device')
        s0 = ctypes.c_uint64(seed0)
        groups = range(self.launches_per_step)
        for i in range(int(k)):
            for g in groups:
                rc = self.lib.cc4_random_actions_group_device(self._h, g, s0, ctypes.c_uint32(t0 + i)) or self.lib.cc4_step_group_device(self._h, g, p, None)
                if rc:
                    self._chk(rc, 'cc4_step_group_device')
        return 0.0

    def run_rollout(self, k, policy='random', seed0=0, t0=0, native=False):
        """A k-step rollout with the policy in the loop and ONE launch of the step engine (cc4_rollout_begin .. cc4_rollout_end, include/cc4.h): per step
        and policy group the gate on the last step's observations, the stand-in policy ('random': the draws of run_random_steps / run_policy_steps;
        'hash': action indices computed from each episode's packed observation row of the step before), the publish -- all enqueued behind the launch
        on the handle's policy stream.  A policy of the caller's takes the stand-in's place between cc4_rollout_wait_obs and cc4_rollout_publish."""
        lib, h = self.lib, self._h
        if native:       # the same sequence enqueued by the library itself (cc4_rollout_standin): no Python between the stream operations
            self._chk(lib.cc4_rollout_standin(h, int(k), 0 if policy == 'random' else 1, ctypes.c_uint64(seed0), ctypes.c_uint32(t0)), 'cc4_rollout_standin')
            return 0.0
        self._chk(lib.cc4_rollout_begin(h, int(k)), 'cc4_rollout_begin')
        G, blk = ctypes.c_int32(), ctypes.c_int32()
        rc = lib.cc4_rollout_groups(h, ctypes.byref(G), ctypes.byref(blk))
        s0 = ctypes.c_uint64(seed0)
        for j in range(int(k)):
            for g in range(G.value):
                # on the group's own policy stream: one operation for the publish of its last pass and the gate of this one (cc4_rollout_sync), then the policy
                rc = rc or lib.cc4_rollout_sync(h, g if j > 0 else -1, j - 1, g, j, None)
                rc = rc or (lib.cc4_rollout_random_policy(h, g, j, s0, ctypes.c_uint32(t0 + j), None) if policy == 'random' else lib.cc4_rollout_hash_policy(h, g, j, None))
        for g in range(G.value):
            rc = rc or lib.cc4_rollout_sync(h, g, int(k) - 1, -1, 0, None)
        end = lib.cc4_rollout_end(h)
        self._chk(rc or end, 'cc4_rollout')
        return 0.0

    def rollout_obs_packed(self, j):
        """Host copy of the packed observation rows the policy of step j of the last rollout read ([N, 148] bytes, 2 bits per value: the observations
        after step j - 1; the ring keeps the last 32 steps)."""
        p = ctypes.c_void_p()
        self._chk(self.lib.cc4_rollout_obs_packed(self._h, int(j), ctypes.byref(p)), 'cc4_rollout_obs_packed')
        out = np.zeros((self.num_envs, 148), np.uint8)
        self._chk(self.lib.cc4_debug_copy_from_device(self._h, out.ctypes.data_as(ctypes.c_void_p), p, out.nbytes), 'cc4_debug_copy_from_device')
        return out

    def rollout_actions(self, j):
        """Host copy of the action slot step j of the last rollout read ([N, 5]; slots alternate: valid for its last two steps)."""
        p = ctypes.c_void_p()
        self._chk(self.lib.cc4_rollout_actions(self._h, int(j), ctypes.byref(p)), 'cc4_rollout_actions')
        out = np.zeros((self.num_envs, L.NUM_BLUE), np.int32)
        self._chk(self.lib.cc4_debug_copy_from_device(self._h, out.ctypes.data_as(ctypes.c_void_p), p, out.nbytes), 'cc4_debug_copy_from_device')
        return out

    def device_actions(self):
        """cc4_get_actions: host copy of the handle's device action buffer ([N, 5]; after run_random_steps: the indices the last
        step drew in-kernel)."""
        out = np.zeros((self.num_envs, L.NUM_BLUE), np.int32)
        self._chk(self.lib.cc4_get_actions(self._h, out.ctypes.data_as(ctypes.c_void_p)), 'cc4_get_actions')
        return out

    def synchronize(self):
        self._chk(self.lib.cc4_synchronize(self._h), 'cc4_synchronize')


def split_obs(obs):
    """[N,578] -> list of 5 arrays [N,92|210] (agent order blue_agent_0..4)."""
    return [obs[:, o:o + n] for o, n in zip(L.OBS_OFF, L.OBS_LEN)]


def split_mask(mask):
    return [mask[:, o:o + n] for o, n in zip(L.ACT_OFF, L.ACT_LEN)]


def shard_range(num_envs, rank, world):
    """Static env sharding for multi-GPU runs: contiguous ranges [lo, hi)."""
    per = num_envs // world
    rem = num_envs % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)
