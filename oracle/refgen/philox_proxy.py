"""A numpy-Generator stand-in that lets the REFERENCE simulator (CybORG v4, /root/reference) step under the engine's counter-based
streams, so that the counter (Philox) mode -- the mode bench.py times -- can be pinned to reference trajectories.

TEST INFRASTRUCTURE, build container only (the reference never travels to the GPU box).

How it is used.  The reference takes a ready generator object as its seed (CybORG/env.py:73-77; precedent:
CybORG/Tests/utils.py:81-174 CustomGenerator) and hands that one object to the controller, the state, every host and every
agent (SimulationController.py:105,1041, State.py:77, EnterpriseScenarioGenerator.py:148,527,741,811), and every draw of the
step path is one of four methods on it.  PhiloxProxy
  * delegates to a real Generator(PCG64(SeedSequence(seed))) until it is armed -- so the scenario (CybORG.__init__ and
    reset(seed=None)) is the one the engine generates from the same seed in its numpy-stream mode;
  * once armed, serves every call from the engine's counter streams: Philox4x32-10, key = the 64-bit key, counter =
    (block number, stream id, step, episode) with one stream per (agent, phase) -- csrc/cc4_rng.h ST_* -- identified from the
    CALL SITE: the proxy walks the Python stack for the executing Action object (its class and `agent` give phase and
    agent) or the agent object whose get_action is running.  The distributions are numpy's own algorithms on those words,
    as csrc/cc4_rng.h restates them for this mode: bounded integers = buffered Lemire on 32-bit words (a one-option range
    draws nothing), choice(p) = cdf.searchsorted(random(), 'right'), random() = one 32-bit word * 2^-32 (every threshold on
    the path is a multiple of 1/100 or 1/4; cc4_rng.h rng_random), choice(replace=False) of one = one bounded draw.
  * r04, the SCENARIO GENERATION as well (arm(..., generation=True) before CybORG(...) is built): create_scenario and
    State.__init__ draw from the engine's generation streams (cc4_engine.h env_reset_counter_mode), again told apart by call
    site -- _generate_subnet / _generate_hosts / _generate_blue_agents / _generate_red_agents draw from the main reset stream
    (ST_RESET) in the order they run; everything under _generate_linux_host(hostname, ...) from that host's stream
    (ST_GEN_HOST + host id), except the second and later iterations of _generate_pid's retry loop, which take the host's
    re-draw stream (ST_GEN_REDRAW + host id); Host.create_pid under State.__init__ from ST_GEN_SESS + host id.  All of them
    with the reset step word 0xFFFFFFFF and the episode word the engine bumps at every reset: begin_reset() before
    CybORG(...) and before every reset().
  * The two draws that exist only for numpy-stream parity are not made in the counter mode and are not
    served from any stream here either: the action-order shuffle (SimulationController.py:418: the order is never used) leaves
    the list alone, Host.get_ephemeral_port (Host.py:175-187: the value is unobservable on this path) gets ports from a
    private sequence that never collides.
"""
import sys
import numpy as np

M32 = 0xFFFFFFFF
ST_BLUE_EXE, ST_GREEN_POL, ST_GREEN_EXE, ST_GREEN_PHISH = 0x100, 0x200, 0x300, 0x400
ST_RED_POL, ST_RED_EXE, ST_RED_RSC, ST_BLUE_POL = 0x500, 0x600, 0x700, 0xB00
ST_RESET, ST_GEN_HOST, ST_GEN_REDRAW, ST_GEN_SESS = 0x0, 0x800, 0x900, 0xA00
RESET_STEP_WORD = 0xFFFFFFFF      # the step word of every generation stream (cc4_rng.h rng_begin_episode)
GEN_SUBNETS = ('restricted_zone_a_subnet', 'operational_zone_a_subnet', 'restricted_zone_b_subnet', 'operational_zone_b_subnet',
               'contractor_network_subnet', 'public_access_zone_subnet', 'admin_network_subnet', 'office_network_subnet', 'internet_subnet')


def gen_host_index(hostname):
    """hostname -> the engine's host id (subnet * 17 + slot; csrc/cc4_state.h)."""
    hostname = str(hostname)
    if hostname == 'root_internet_host_0':
        return 136
    for s, sn in enumerate(GEN_SUBNETS):
        if hostname.startswith(sn + '_'):
            rest = hostname[len(sn) + 1:]
            if rest == 'router':
                return s * 17
            if rest.startswith('user_host_'):
                return s * 17 + 1 + int(rest[len('user_host_'):])
            if rest.startswith('server_host_'):
                return s * 17 + 11 + int(rest[len('server_host_'):])
    raise ValueError(hostname)


def philox4x32_10(c, k0, k1):
    """Random123 Philox4x32-10 (csrc/cc4_rng.h philox4x32_10)."""
    c = list(c)
    for _ in range(10):
        p0 = 0xD2511F53 * c[0]
        p1 = 0xCD9E8D57 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & M32, p1 & M32, ((p0 >> 32) ^ c[3] ^ k1) & M32, p0 & M32]
        k0 = (k0 + 0x9E3779B9) & M32
        k1 = (k1 + 0xBB67AE85) & M32
    return c


class PhiloxProxy:
    def __init__(self, seed, key=None, episode=1):
        """seed: the numpy seed of the scenario stream; key: Philox key of the dynamics (default: seed); episode: the 4th
        counter word (cc4_set_seed / cc4o_set_seed leave it at 1: rng_seed + rng_begin_episode)."""
        self._pcg = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        self.key = int(seed if key is None else key)
        self.episode = int(episode)
        self.armed = False
        self.step = 0
        self._words = {}        # stream id -> 32-bit words handed out this step
        self._blocks = {}       # (stream id, block number) -> 4 words
        self._port = 0
        self.calls = {}         # stream class -> number of calls served (diagnostics)
        self._action_cls = None

    # ---- life cycle
    def arm(self, action_base_class, generation=False):
        """From now on: counter streams.  action_base_class: CybORG.Simulator.Actions.Action (to recognise executing actions).
        generation: the scenario generation draws from the counter streams too (begin_reset() before every CybORG(...) / reset())."""
        self._action_cls = action_base_class
        self.armed = True
        if generation:
            self.episode = 0                  # rng_seed leaves the episode word 0; every reset bumps it (rng_begin_episode)

    def begin_reset(self):
        """Call before CybORG(...) is constructed and before every reset(): the engine's reset bumps the episode word and draws with
        the reset step word."""
        self.episode += 1
        self.step = RESET_STEP_WORD
        self._words.clear()
        self._blocks.clear()

    def begin_step(self, step):
        """Call before every CybORG step with the step number the engine's row holds (EnvState.step_count)."""
        self.step = int(step)
        self._words.clear()
        self._blocks.clear()

    @property
    def bit_generator(self):       # the scenario stream's position, for the PCG part of the comparison
        return self._pcg.bit_generator

    # ---- call-site -> stream
    def _site(self):
        f = sys._getframe(2)
        outer_action = None
        classes = []
        policy_self = None
        fn_names = []
        while f is not None:
            name = f.f_code.co_name
            fn_names.append(name)
            slf = f.f_locals.get('self')
            if name == 'get_ephemeral_port':
                return 'port', None
            if slf is not None and slf is not self:
                if isinstance(slf, self._action_cls):
                    outer_action = slf
                    classes.append(type(slf).__name__)
                elif name == 'get_action' and hasattr(slf, 'np_random') and policy_self is None and outer_action is None:
                    policy_self = slf
            f = f.f_back
        if outer_action is not None:
            agent = outer_action.agent
            idx = int(agent.rsplit('_', 1)[1])
            if 'PhishingEmail' in classes:
                # the outermost action is the GreenLocalWork of the green agent that sent the email
                return 'phish', ST_GREEN_PHISH + idx
            if 'RedSessionCheck' in classes:
                return 'rsc', ST_RED_RSC + idx
            if agent.startswith('blue'):
                return 'blue_exe', ST_BLUE_EXE + idx
            if agent.startswith('green'):
                return 'green_exe', ST_GREEN_EXE + idx
            if agent.startswith('red'):
                return 'red_exe', ST_RED_EXE + idx
            raise RuntimeError(f'PhiloxProxy: executing action of unknown agent {agent}')
        if policy_self is not None:
            nm = getattr(policy_self, 'name', None)
            if nm is None:
                raise RuntimeError(f'PhiloxProxy: agent object {type(policy_self).__name__} has no name')
            idx = int(nm.rsplit('_', 1)[1])
            if nm.startswith('green'):
                return 'green_pol', ST_GREEN_POL + idx
            if nm.startswith('red'):
                return 'red_pol', ST_RED_POL + idx
            if nm.startswith('blue'):
                return 'blue_pol', ST_BLUE_POL + idx
        if 'sort_action_order' in fn_names:
            return 'shuffle', None
        # ---- scenario generation (create_scenario, State.__init__)
        if '_generate_linux_host' in fn_names:
            f = sys._getframe(2)
            host, redraw = None, False
            while f is not None:
                if f.f_code.co_name == '_generate_linux_host':
                    host = gen_host_index(f.f_locals['hostname'])
                elif f.f_code.co_name == '_generate_pid':
                    redraw = 'pid' in f.f_locals          # the loop has been round once already: this is a retry
                f = f.f_back
            return ('gen_redraw', ST_GEN_REDRAW + host) if redraw else ('gen_host', ST_GEN_HOST + host)
        if fn_names[0] == 'create_pid' and 'add_session' in fn_names:
            host_obj = sys._getframe(2).f_locals['self']
            return 'gen_sess', ST_GEN_SESS + gen_host_index(host_obj.hostname)
        for nm in ('_generate_subnet', '_generate_hosts', '_generate_blue_agents', '_generate_red_agents'):
            if nm in fn_names:
                return 'gen_main', ST_RESET
        raise RuntimeError('PhiloxProxy: draw from an unrecognised call site: ' + ' <- '.join(fn_names[:8]))

    # ---- words of a stream (cc4_rng.h rng_next32 in mode 1: the four words of block 0, then of block 1, ...)
    def _next32(self, stream):
        w = self._words.get(stream, 0)
        self._words[stream] = w + 1
        blk = w >> 2
        key = (stream, blk)
        if key not in self._blocks:
            self._blocks[key] = philox4x32_10([blk, stream, self.step & M32, self.episode & M32], self.key & M32, (self.key >> 32) & M32)
        return self._blocks[key][w & 3]

    def _below(self, stream, n):
        """cc4_rng.h rng_below == numpy buffered_bounded_lemire_uint32 with rng = n - 1; n <= 1 draws nothing."""
        if n <= 1:
            return 0
        m = self._next32(stream) * n
        leftover = m & M32
        if leftover < n:
            threshold = (M32 - (n - 1)) % n
            while leftover < threshold:
                m = self._next32(stream) * n
                leftover = m & M32
        return m >> 32

    def _random(self, stream):
        return float(self._next32(stream)) * (1.0 / 4294967296.0)

    def _count(self, kind):
        self.calls[kind] = self.calls.get(kind, 0) + 1

    # ---- the Generator surface the reference uses
    def integers(self, low, high=None, size=None, dtype=np.int64, endpoint=False):
        if not self.armed:
            return self._pcg.integers(low, high, size=size, dtype=dtype, endpoint=endpoint)
        assert size is None
        kind, stream = self._site()
        self._count(kind)
        if high is None:
            low, high = 0, low
        if endpoint:
            high = high + 1
        if kind == 'port':
            assert (low, high) == (49152, 60000)
            p = 49152 + self._port % (60000 - 49152)
            self._port += 1
            return np.int64(p)
        assert stream is not None, kind
        return np.int64(low + self._below(stream, int(high) - int(low)))

    def choice(self, a, size=None, replace=True, p=None, axis=0, shuffle=True):
        if not self.armed:
            return self._pcg.choice(a, size=size, replace=replace, p=p, axis=axis, shuffle=shuffle)
        assert size is None and axis == 0
        kind, stream = self._site()
        self._count(kind)
        assert stream is not None, kind
        if isinstance(a, (int, np.integer)):
            arr, n = None, int(a)
        else:
            arr = np.asarray(a)
            n = arr.shape[0]
        if n == 0:
            raise ValueError("a cannot be empty unless no samples are taken")
        if p is not None:
            assert replace
            pp = np.asarray(p, dtype=np.float64)
            cdf = pp.cumsum()
            cdf /= cdf[-1]
            idx = int(cdf.searchsorted(self._random(stream), side='right'))
        else:
            idx = self._below(stream, n)      # replace=False with one sample is one bounded draw too (Floyd's loop runs once)
        if arr is None:
            return np.int64(idx)
        return arr[idx]

    def random(self, size=None, dtype=np.float64, out=None):
        if not self.armed:
            return self._pcg.random(size=size, dtype=dtype, out=out)
        assert size is None
        kind, stream = self._site()
        self._count(kind)
        assert stream is not None, kind
        return self._random(stream)

    def shuffle(self, x, axis=0):
        if not self.armed:
            return self._pcg.shuffle(x, axis=axis)
        kind, _ = self._site()
        self._count(kind)
        assert kind == 'shuffle', kind      # only the action-order shuffle exists on the path; its result is never used
        return None

    def __getattr__(self, name):
        raise AttributeError(f'PhiloxProxy: the reference asked for Generator.{name}, which the step path was not known to use')
