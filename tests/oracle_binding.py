"""ctypes binding of oracle/liboracle.so -- the CPU oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.  The product package
(cage_challenge_4_amd) never does."""
import ctypes
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, 'oracle')
LIB = os.path.join(ORACLE_DIR, 'liboracle.so')


def build():
    subprocess.check_call(['make', '-C', ORACLE_DIR, '-s'])


def usable_cores():
    """Hardware threads this process may really use: affinity mask and the container's CPU quota (cgroup v2 cpu.max).  OpenMP's default is every core it
    can see; on a box whose cgroup grants 16 of 128, the other 112 threads only spin (the GPU suite waits for the oracle most of its time)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    try:
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


_LIB = None


def load():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB):
        build()
    lib = ctypes.CDLL(LIB)
    lib.cc4o_set_threads.argtypes = [ctypes.c_int]
    if not os.environ.get('OMP_NUM_THREADS'):
        lib.cc4o_set_threads(usable_cores())
    lib.cc4o_create.restype = ctypes.c_void_p
    lib.cc4o_create.argtypes = [ctypes.c_int]
    lib.cc4o_create2.restype = ctypes.c_void_p
    lib.cc4o_create2.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.cc4o_cold_bytes.restype = ctypes.c_size_t
    lib.cc4o_cold_bytes.argtypes = [ctypes.c_void_p]
    lib.cc4o_cold_ptr.restype = ctypes.c_void_p
    lib.cc4o_cold_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cc4o_destroy.argtypes = [ctypes.c_void_p]
    lib.cc4o_reset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.cc4o_step.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    lib.cc4o_step_all.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.cc4o_step_check_marks.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.cc4o_step_batch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    lib.cc4o_obs.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.cc4o_reward.restype = ctypes.c_float
    lib.cc4o_reward.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cc4o_done.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cc4o_err.restype = ctypes.c_uint32
    lib.cc4o_err.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cc4o_mask.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.cc4o_rng_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.cc4o_state_bytes.restype = ctypes.c_size_t
    lib.cc4o_state_ptr.restype = ctypes.c_void_p
    lib.cc4o_state_ptr.argtypes = [ctypes.c_void_p, ctypes.c_int]
    lib.cc4o_topology.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.cc4o_dump.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    lib.cc4o_true_state.restype = ctypes.c_longlong
    lib.cc4o_true_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_size_t]
    _LIB = lib
    return lib


AGENT_ACTION_DTYPE = [('type', 'i1'), ('host', 'u1'), ('arg', 'u1'), ('ticks', 'u1'), ('session', 'u2'), ('flags', 'u1'), ('pad', 'u1'),
                      ('rate0', 'f8'), ('rate1', 'f8')]          # ExtAct (csrc/cc4_state.h) == cc4_agent_action (include/cc4.h)


class OracleVecEnv:
    """Same call shape as cage_challenge_4_amd.CC4VecEnv, stepping episodes serially on the host."""
    def __init__(self, num_envs, steps=500, rng_mode=0, autoreset=False, device_id=0, red_policy=0, green_policy=0, topology_seed=0,
                 strict=True, blue_policy=0):
        self.lib = load()
        self.policy = (red_policy & 3) | (0x10 if green_policy else 0) | (0x40 if green_policy == 2 else 0) | (0x20 if blue_policy else 0)
        self.num_envs = num_envs
        self.steps = steps
        self.rng_mode = rng_mode
        self.autoreset = autoreset
        self._h = ctypes.c_void_p(self.lib.cc4o_create2(num_envs, int(steps)))   # the cold containers are sized from the episode length
        self.lib.cc4o_set_topology_seed.argtypes = [ctypes.c_void_p, ctypes.c_uint32]
        self.lib.cc4o_set_topology_seed(self._h, int(topology_seed))
        self._obs = np.zeros((num_envs, 578), np.int32)
        self._rew = np.zeros(num_envs, np.float32)
        self._done = np.zeros(num_envs, bool)
        self._err = np.zeros(num_envs, np.uint32)

    def close(self):
        if self._h:
            self.lib.cc4o_destroy(self._h)
            self._h = None

    def _collect(self, i, reward=True):
        self.lib.cc4o_obs(self._h, i, self._obs[i].ctypes.data_as(ctypes.c_void_p))
        self._rew[i] = self.lib.cc4o_reward(self._h, i) if reward else 0.0
        self._done[i] = bool(self.lib.cc4o_done(self._h, i))
        self._err[i] = self.lib.cc4o_err(self._h, i)

    def reset(self, seeds=None, env_mask=None):
        if seeds is not None and np.isscalar(seeds):
            seeds = np.uint64(seeds) + np.arange(self.num_envs, dtype=np.uint64)
        for i in range(self.num_envs):
            if env_mask is not None and not env_mask[i]:
                continue
            if seeds is None:
                self.lib.cc4o_reset(self._h, i, 0, self.rng_mode, self.steps, 1, self.policy)
            else:
                self.lib.cc4o_reset(self._h, i, ctypes.c_uint64(int(seeds[i])), self.rng_mode, self.steps, 0, self.policy)
            self._collect(i, reward=False)
        return self._obs

    def step(self, actions=None, messages=None):
        for i in range(self.num_envs):
            if self.autoreset and self._done[i]:
                self.lib.cc4o_reset(self._h, i, 0, self.rng_mode, self.steps, 1, self.policy)
                self._collect(i, reward=False)
                continue
            a = None if actions is None else np.ascontiguousarray(actions[i], np.int32)
            m = None if messages is None else np.ascontiguousarray(messages[i], np.uint8)
            ap = None if a is None else a.ctypes.data_as(ctypes.c_void_p)
            mp = None if m is None else m.ctypes.data_as(ctypes.c_void_p)
            if getattr(self, 'check_marks', False):   # the engine's dirty-row marks (StepWork.hdirty) cover every HostDyn row the step changed
                mk = ctypes.c_int(0)
                self.unmarked_rows = getattr(self, 'unmarked_rows', 0) + self.lib.cc4o_step_check_marks(self._h, i, ap, mp, ctypes.byref(mk))
                self.marked_rows = getattr(self, 'marked_rows', 0) + mk.value
            else:
                self.lib.cc4o_step(self._h, i, ap, mp)
            self._collect(i)
        return self._obs, self._rew, self._done, {'err': self._err}

    def agent_actions(self, kind):
        a = np.zeros((self.num_envs, 6 if kind == 'red' else 80), dtype=AGENT_ACTION_DTYPE)
        a['type'] = -1
        return a

    def step_ex(self, actions=None, messages=None, red=None, green=None):
        """Same call shape as CC4VecEnv.step_ex (cc4_step_ex): the step with submitted red / green actions (cc4o_step_ex)."""
        self.lib.cc4o_step_ex.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        for i in range(self.num_envs):
            a = None if actions is None else np.ascontiguousarray(actions[i], np.int32)
            m = None if messages is None else np.ascontiguousarray(messages[i], np.uint8)
            ext = np.zeros(86, dtype=AGENT_ACTION_DTYPE)
            ext['type'] = -1
            if red is not None:
                ext[:6] = red[i]
            if green is not None:
                ext[6:] = green[i]
            self.lib.cc4o_step_ex(self._h, i, None if a is None else a.ctypes.data_as(ctypes.c_void_p),
                                  None if m is None else m.ctypes.data_as(ctypes.c_void_p), ext.ctypes.data_as(ctypes.c_void_p))
            self._collect(i)
        return self._obs, self._rew, self._done, {'err': self._err}

    def edit_state(self, env, op, a0=0, a1=0, a2=0):
        self.lib.cc4o_edit_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 5
        rc = int(self.lib.cc4o_edit_state(self._h, int(env), int(op), int(a0), int(a1), int(a2)))
        if rc < 0:
            raise RuntimeError(f'cc4o_edit_state({op}, {a0}, {a1}, {a2}) failed')
        return rc

    def step_batch(self, actions=None, messages=None):
        """step() for the whole batch in one native call (cc4o_step_batch: OpenMP over episodes, incl. the autoreset): same
        results, seconds instead of minutes at 8192 episodes."""
        a = None if actions is None else np.ascontiguousarray(actions, np.int32)
        m = None if messages is None else np.ascontiguousarray(messages, np.uint8)
        done8 = np.zeros(self.num_envs, np.uint8)
        P = lambda v: None if v is None else v.ctypes.data_as(ctypes.c_void_p)   # noqa: E731
        self.lib.cc4o_step_batch(self._h, P(a), P(m), int(bool(self.autoreset)), self.rng_mode, self.steps, self.policy,
                                 P(self._obs), P(self._rew), P(done8), P(self._err))
        self._done[:] = done8.astype(bool)
        return self._obs, self._rew, self._done, {'err': self._err}

    def reset_batch(self, seed0):
        """reset(seeds=seed0 + i) without the per-episode observation / mask fetches of reset() (first observations come with
        state_obs())."""
        for i in range(self.num_envs):
            self.lib.cc4o_reset(self._h, i, ctypes.c_uint64(int(seed0) + i), self.rng_mode, self.steps, 0, self.policy)
            self._collect(i, reward=False)
        return self._obs

    def set_seed(self, seeds):
        if np.isscalar(seeds):
            seeds = np.uint64(seeds) + np.arange(self.num_envs, dtype=np.uint64)
        self.lib.cc4o_set_seed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]
        for i in range(self.num_envs):
            self.lib.cc4o_set_seed(self._h, i, ctypes.c_uint64(int(seeds[i])), self.rng_mode)

    def set_generators(self, generators):
        from cage_challenge_4_amd.vec_env import pcg64_words
        self.lib.cc4o_set_rng_state.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        for i, g in enumerate(generators):
            w = np.array(pcg64_words(g), np.uint64)
            self.lib.cc4o_set_rng_state(self._h, i, w.ctypes.data_as(ctypes.c_void_p))

    @property
    def action_mask(self):
        return self.mask()

    def topology(self, i=0):
        buf = np.zeros(27 + 2 * 137, np.uint8)
        self.lib.cc4o_topology(self._h, i, buf.ctypes.data_as(ctypes.c_void_p))
        return buf

    def mask(self):
        m = np.zeros((self.num_envs, 570), np.uint8)
        for i in range(self.num_envs):
            self.lib.cc4o_mask(self._h, i, m[i].ctypes.data_as(ctypes.c_void_p))
        return m.astype(bool)

    def rng_state(self):
        out = np.zeros((self.num_envs, 7), np.uint64)
        for i in range(self.num_envs):
            self.lib.cc4o_rng_state(self._h, i, out[i].ctypes.data_as(ctypes.c_void_p))
        return out

    def get_state(self, i):
        n = self.lib.cc4o_state_bytes()
        p = self.lib.cc4o_state_ptr(self._h, i)
        return np.frombuffer((ctypes.c_uint8 * n).from_address(p), np.uint8).copy()

    def get_cold(self, i):
        nc = self.lib.cc4o_cold_bytes(self._h)
        return np.frombuffer((ctypes.c_uint8 * nc).from_address(self.lib.cc4o_cold_ptr(self._h, i)), np.uint8).copy()

    def snapshot(self, i):
        """(hot row, cold row) of episode i as byte arrays: the layout cc4_get_state / cc4_get_cold return."""
        nc = self.lib.cc4o_cold_bytes(self._h)
        cold = np.frombuffer((ctypes.c_uint8 * nc).from_address(self.lib.cc4o_cold_ptr(self._h, i)), np.uint8).copy()
        return self.get_state(i), cold

    def restore(self, i, snap):
        hot, cold = snap
        hot = np.ascontiguousarray(hot, np.uint8); cold = np.ascontiguousarray(cold, np.uint8)
        assert hot.size == self.lib.cc4o_state_bytes() and cold.size == self.lib.cc4o_cold_bytes(self._h)
        ctypes.memmove(self.lib.cc4o_state_ptr(self._h, i), hot.ctypes.data, hot.size)
        ctypes.memmove(self.lib.cc4o_cold_ptr(self._h, i), cold.ctypes.data, cold.size)
        self._done[i] = bool(self.lib.cc4o_done(self._h, i))

    def enable_event_log(self, on=True):
        self.lib.cc4o_enable_event_log.argtypes = [ctypes.c_void_p, ctypes.c_int]
        self.lib.cc4o_enable_event_log(self._h, int(bool(on)))

    def true_state_json(self, i=0):
        need = int(self.lib.cc4o_true_state(self._h, i, None, 0))
        buf = ctypes.create_string_buffer(need)
        self.lib.cc4o_true_state(self._h, i, buf, need)
        return buf.value.decode()

    def dump(self, i):
        buf = ctypes.create_string_buffer(1 << 20)
        n = self.lib.cc4o_dump(self._h, i, buf, len(buf))
        return buf.raw[:n].decode()


def random_actions(seed0, t, num_envs):
    """Host restatement of k_random_actions (csrc/cc4_k_misc.hip): Philox4x32-10 key (seed0+env), counter (t, agent, 0xB10E, 0)."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    e = np.arange(num_envs, dtype=np.uint64)[:, None] + np.uint64(seed0)
    b = np.arange(5, dtype=np.uint64)[None, :]
    c = [np.full((num_envs, 5), t, np.uint64), np.broadcast_to(b, (num_envs, 5)).copy(),
         np.full((num_envs, 5), 0xB10E, np.uint64), np.zeros((num_envs, 5), np.uint64)]
    k0 = np.broadcast_to(e & np.uint64(0xFFFFFFFF), (num_envs, 5)).copy()
    k1 = np.broadcast_to(e >> np.uint64(32), (num_envs, 5)).copy()
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        n0 = ((p1 >> np.uint64(32)) ^ c[1] ^ k0) & mask
        n1 = p1 & mask
        n2 = ((p0 >> np.uint64(32)) ^ c[3] ^ k1) & mask
        n3 = p0 & mask
        c = [n0, n1, n2, n3]
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    rng = np.array([82, 82, 82, 82, 242], np.uint64)[None, :]
    return ((c[0] * rng) >> np.uint64(32)).astype(np.int32)
