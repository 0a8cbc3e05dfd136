"""Drop-in single-episode surface: the reference's class names, argument meaning and error behaviour for the
hot path, backed by the HIP engine (an N=1 view of CC4VecEnv).

    from cage_challenge_4_amd import (CybORG, EnterpriseScenarioGenerator, SleepAgent, EnterpriseGreenAgent,
                                      FiniteStateRedAgent, BlueFlatWrapper, BlueEnterpriseWrapper, EnterpriseMAE)
    sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent,
                                     red_agent_class=FiniteStateRedAgent, steps=500)
    env = BlueFlatWrapper(CybORG(sg, seed=123))
    obs, info = env.reset()
    obs, rew, term, trunc, info = env.step({'blue_agent_0': 3, ...})

Mirrors: CybORG/env.py:53-77,218-243; CybORG/Simulator/Scenarios/EnterpriseScenarioGenerator.py:87-121;
CybORG/Agents/Wrappers/BlueFixedActionWrapper.py:26-379; BlueFlatWrapper.py:31-322; BlueEnterpriseWrapper.py:25-122;
EnterpriseMAE.py:10-72.  Outputs are bit-identical to the reference under the same seed (rng_mode PCG64)."""
import functools
import json
import os
import ctypes
import numpy as np
from . import _lib as L
from .vec_env import CC4VecEnv, RNG_PCG64, split_obs, split_mask, raise_on_engine_error
from .spaces import Discrete, MultiDiscrete, MultiBinary
from . import actions as A

SUBNET_NAMES = ['restricted_zone_a_subnet', 'operational_zone_a_subnet', 'restricted_zone_b_subnet',
                'operational_zone_b_subnet', 'contractor_network_subnet', 'public_access_zone_subnet',
                'admin_network_subnet', 'office_network_subnet', 'internet_subnet']   # SUBNET enum order (ESG.py:40-51)
BLUE_SUBNETS = [['restricted_zone_a_subnet'], ['operational_zone_a_subnet'], ['restricted_zone_b_subnet'],
                ['operational_zone_b_subnet'],
                ['public_access_zone_subnet', 'admin_network_subnet', 'office_network_subnet']]  # ESG.py:643-649
json_loads = json.loads
MAX_USER_HOSTS, MAX_SERVER_HOSTS = 10, 6
NUM_MESSAGES, MESSAGE_LENGTH = 4, 8
EMPTY_MESSAGE = np.zeros(MESSAGE_LENGTH, dtype=bool)


# ---- agent-class markers (the engine implements exactly these built-in policies)
class SleepAgent:
    """CybORG/Agents/SimpleAgents/ConstantAgent.py:32-36"""
    def __init__(self, name=None, **kwargs):
        self.name = name


class EnterpriseGreenAgent:
    """CybORG/Agents/SimpleAgents/EnterpriseGreenAgent.py:8-85 (fp_detection_rate = phishing_error_rate = 0.01)"""


class FiniteStateRedAgent:
    """CybORG/Agents/SimpleAgents/FiniteStateRedAgent.py:15-550"""


class RandomSelectRedAgent:
    """CybORG/Agents/SimpleAgents/RandomSelectRedAgent.py:8-148 (uniform command + uniform known parameters, incl. Withdraw)"""


class cc4BlueRandomAgent:
    """CybORG/Agents/SimpleAgents/RandomAgent.py:14-69 (epsilon = 1: a uniform action class out of Monitor, Analyse, Restore,
    Remove, DeployDecoy, Sleep, then a uniform host of the agent's zone, routers included); as `blue_agent_class` it acts for
    every blue agent a step gets no action for"""
    def __init__(self, name=None, **kwargs):
        self.name = name


class DiscoveryFSRed(FiniteStateRedAgent):
    """CybORG/Agents/SimpleAgents/FSMRedVariants.py:80-122 (host-state priorities, prioritise_servers, own probability matrix)"""


class EnterpriseScenarioGenerator:
    """Configuration holder with the reference's constructor (ESG.py:95-121) and class constants (:87-93)."""
    MIN_USER_HOSTS = 3
    MAX_USER_HOSTS = 10
    MIN_SERVER_HOSTS = 1
    MAX_SERVER_HOSTS = 6
    MAX_ADDON_SERVICES = 10
    MAX_BANDWIDTH = 100
    MESSAGE_LENGTH = 8

    def __init__(self, blue_agent_class=None, red_agent_class=None, green_agent_class=None, steps: int = 100, host_agents=False):
        # The engine runs these policies on the device, selected by CLASS NAME: the reference's own classes
        # (CybORG.Agents.FiniteStateRedAgent, ...) and this package's markers select the same device policy, bit-exact under the seed.
        # A class of ANOTHER name is instantiated on the host, one object per agent, as the reference does (ESG.py:95-121, :693-696,
        # :736-748, :805-817): the engine's own policy for that team then sleeps, the object's get_action(observation, action_space) is
        # called every step with the agent's dict observation, and what it returns is submitted through cc4_step_ex (the slow path: one
        # host round trip per step -- what a scripted or learning red agent costs in the reference too).  host_agents=True (or a class
        # attribute cc4_host_agent = True) sends a class to the host path although its name is a built-in's.  The name selects the device
        # policy only for classes from the `CybORG` package or this one: a class of any other module whose name collides with a built-in
        # (a user's modified FiniteStateRedAgent) runs ITS get_action on the host, unless it sets cc4_device_policy = True.
        blue = {'SleepAgent': 0, 'cc4BlueRandomAgent': 1}
        red = {'FiniteStateRedAgent': 0, 'SleepAgent': 1, 'DiscoveryFSRed': 2, 'RandomSelectRedAgent': 3}
        green = {'EnterpriseGreenAgent': 0, 'SleepAgent': 1}
        # None -> SleepAgent, as the reference's generators do (ESG.py:745-748, :815-817)
        bn = 'SleepAgent' if blue_agent_class is None else getattr(blue_agent_class, '__name__', None)
        rn = 'SleepAgent' if red_agent_class is None else getattr(red_agent_class, '__name__', None)
        gn = 'SleepAgent' if green_agent_class is None else getattr(green_agent_class, '__name__', None)
        self.custom = {}                                   # team -> class whose objects act from the host
        for team, name, table, cls in (('blue', bn, blue, blue_agent_class), ('red', rn, red, red_agent_class), ('green', gn, green, green_agent_class)):
            builtin = name in table and not (host_agents and cls is not None and name != 'SleepAgent') and not getattr(cls, 'cc4_host_agent', False)
            mod = getattr(cls, '__module__', '') or ''
            ours = cls is None or mod == 'CybORG' or mod.startswith(('CybORG.', 'cage_challenge_4_amd'))
            if builtin and not ours and name != 'SleepAgent':
                # a class of ANOTHER package whose name merely collides with a built-in (a user's modified FiniteStateRedAgent): its own
                # get_action must run -- host path by default; cc4_device_policy = True on the class opts into the device policy by name
                if getattr(cls, 'cc4_device_policy', False):
                    pass
                elif hasattr(cls, 'get_action'):
                    builtin = False
            if not builtin:
                if not callable(cls) or not hasattr(cls, 'get_action'):
                    raise TypeError(f'{team}_agent_class={cls!r}: an agent class needs get_action(observation, action_space)')
                self.custom[team] = cls
        self.blue_policy = 0 if 'blue' in self.custom else blue[bn]
        self.red_policy = 1 if 'red' in self.custom else red[rn]
        # (2: no built-in green policy, but -- unlike a SleepAgent's -- an action space that holds the green actions, ESG.py:714,742-746)
        self.green_policy = 2 if 'green' in self.custom else green[gn]
        self.blue_agent_class = blue_agent_class
        self.red_agent_class = red_agent_class
        self.green_agent_class = green_agent_class
        self.steps = int(steps)

    def __str__(self):
        return "EnterpriseScenarioGenerator"


class CybORG:
    """CybORG(scenario_generator, agents=None, seed=None)  (env.py:53-77).  seed: int or None."""
    def __init__(self, scenario_generator, agents=None, seed=None, device_id=0, rng_mode=RNG_PCG64, vec_factory=None):
        assert isinstance(scenario_generator, EnterpriseScenarioGenerator), \
            f'Scenario generator object of type {type(scenario_generator)} must be a subclass of ScenarioGenerator'
        # env.py:53-77: agents = {agent_name: agent object} overrides the scenario's agent for those agents (evaluation.py:69 passes the
        # string "sim" here; it selects nothing).  Such objects act from the host, like the objects of a custom agent class.
        self._agent_overrides = dict(agents) if isinstance(agents, dict) else {}
        self.scenario_generator = scenario_generator
        if seed is None:
            seed = int.from_bytes(os.urandom(8), 'little') >> 1
        generator = None
        if not isinstance(seed, (int, np.integer)):
            # env.py:73-76 also takes a ready generator.  A numpy Generator(PCG64) is adopted (its stream continues on the device;
            # the Python object is not advanced); anything else -- e.g. the AlwaysTrue / AlwaysFalse stubs of CybORG/Tests/utils.py
            # -- is Python code the device cannot call
            if not isinstance(seed, np.random.Generator):
                raise NotImplementedError("only int seeds and numpy Generator(PCG64) objects can drive the device RNG")
            generator, seed = seed, 0
        # vec_factory: anything with CC4VecEnv's call shape (tests inject the CPU oracle; the default is the HIP engine)
        self.vec = (vec_factory or CC4VecEnv)(1, steps=scenario_generator.steps, rng_mode=rng_mode, device_id=device_id,
                                              red_policy=scenario_generator.red_policy,
                                              green_policy=scenario_generator.green_policy,
                                              blue_policy=scenario_generator.blue_policy)
        self.vec.enable_event_log(True)                        # single episode: keep the per-step event detail for get_observation
        self._lazy_log, self._events_fresh = False, False     # (the fixed-action wrappers switch to the log on demand: _event_log_on_demand)
        if generator is not None:
            self.vec.set_generators([generator])
            self.vec.reset(seeds=None)                           # the scenario is drawn from the adopted stream
        else:
            self.vec.reset(seeds=np.array([seed], np.uint64))  # SimulationController.__init__ creates a scenario
        self.agents = [f'blue_agent_{b}' for b in range(5)]
        self._labels = None
        self._seed = int(seed)
        self._make_host_agents()

    def _tsjson(self):
        """The episode's state document.  With the event log on demand (_event_log_on_demand) the last step is first repeated with the
        log on -- once per step, and only when somebody looks."""
        if self._lazy_log and not self._events_fresh:
            self.vec.replay_logged()
            self._events_fresh = True
        return self.vec.true_state_json(0)

    def _event_log_on_demand(self):
        """Called by the fixed-action wrappers: their step() returns flat vectors, which need no event log -- and the logging build of the
        numpy-stream kernel costs a single episode ~30 us per step.  The log goes OFF; the engine keeps the rows of the previous step
        (cc4_keep_previous) and repeats that step with the log on the first time get_observation / get_true_state / parallel_step ... ask
        what happened in it (cc4_replay_logged): the same answers, paid for only when asked.  Not with host-side agent objects (they read
        their dict observation every step) and not on backends without the replay (the CPU oracle of the tests)."""
        if self._lazy_log or self._host_agents or not hasattr(self.vec, 'replay_logged'):
            return
        self.vec.enable_event_log(False)
        self.vec.keep_previous(True)
        self._lazy_log, self._events_fresh = True, False

    def _make_host_agents(self):
        """The agent objects that act from the host: one per agent of a team with a custom class (constructed as the reference's
        generator does: (name) or (name, np_random=...) -- the reference hands every such object ITS shared generator; here each
        gets a numpy Generator of its own, seeded from the episode's seed and the agent's name, because the simulator's stream lives
        on the device), plus the `agents=` overrides.  Called at construction and at every reset: the reference's reset builds a fresh
        scenario, agent objects included (SimulationController.reset -> _create_agents, SC:153-209), after end_episode() on the old
        ones -- so a learning agent that must survive resets is passed through `agents=`, which is kept."""
        sg = self.scenario_generator
        self._host_agents = {}
        if not getattr(sg, 'custom', None) and not self._agent_overrides:
            return                                          # the default: every policy runs on the device -- nothing to build, no state to fetch
        import inspect
        import zlib
        d = json_loads(self._tsjson())
        names = {'blue': list(self.agents_blue), 'green': [f'green_agent_{g}' for g in range(d['n_green'])],
                 'red': [f'red_agent_{r}' for r in range(6)]}
        for team, cls in getattr(sg, 'custom', {}).items():
            params = inspect.signature(cls.__init__).parameters if hasattr(cls, '__init__') else {}
            for name in names[team]:
                kw = {}
                if 'np_random' in params:
                    kw['np_random'] = np.random.default_rng([self._seed & 0xFFFFFFFF, zlib.crc32(name.encode())])
                try:
                    self._host_agents[name] = cls(name, **kw)
                except TypeError:
                    self._host_agents[name] = cls(**kw) if kw else cls()
        self._host_agents.update(self._agent_overrides)

    def reset(self, agent=None, seed=None):
        """env.py:218-243: seed=None keeps the running stream; an int seed starts a fresh Generator."""
        self.vec.reset(seeds=None if seed is None else np.array([seed], np.uint64))
        self._events_fresh = False
        self._labels = None
        if seed is not None:
            self._seed = int(seed)
        for a in self._host_agents.values():          # AgentInterface.reset -> agent.end_episode (Shared/AgentInterface.py:145-160)
            if hasattr(a, 'end_episode'):
                a.end_episode()
        self._make_host_agents()

    @property
    def unwrapped(self):
        return self

    # ---- the raw CybORG surface around the step (env.py:95-161, 202-216, 266-283, 316-372, 405-415)
    def _state(self):
        from .true_state import decode
        return decode(self._tsjson())

    def _action_labels(self):
        """The fixed action lists of the five blue agents (BlueFixedActionWrapper.py:233-309) for this episode's topology."""
        if getattr(self, '_labels', None) is None:
            self._labels = build_action_labels(self.get_cidr_map(), split_mask(self.vec.action_mask))
        return self._labels

    def _blue_code(self, agent, v, labels=None):
        """What cc4_step_ex takes for a blue agent: the index of the action in the agent's fixed list, or a (type, host / subnet
        pair) code for an action object the list has no slot for, plus the object's `duration` when it is not the class's own."""
        b = self.agents_blue.index(agent)
        labels = labels if labels is not None else self._action_labels()[agent]['labels']
        if type(v) is int or isinstance(v, np.integer):    # the hot case: an index into the agent's list (no object, no duration)
            idx = range(len(labels))[int(v)]                # list semantics: a negative index counts from the end, out of range raises IndexError
            return idx if idx < (242 if b == 4 else 82) else labels.index('Sleep')   # a padded slot is an explicit Sleep, not "no action" (-1)
        name = getattr(v, 'name', None) or type(v).__name__
        if name in ('BlockTrafficZone', 'AllowTrafficZone') and not isinstance(v, (int, np.integer)):
            try:
                idx = A.action_index(v, labels)
            except ValueError:
                # any pair of subnets is a valid Block / Allow for any blue agent: neither parameter is an ActionSpace key
                # (SimulationController.py:1094-1096; Tests/test_cc4/test_BlueRewardMachine.py:133-137)
                to, frm = A._subnet_index(v.to_subnet, ()), A._subnet_index(v.from_subnet, ())
                if to is None or frm is None:
                    raise
                idx = A.BLUE_RAW_ACTION | ((6 if name == 'BlockTrafficZone' else 7) << 8) | to | (frm << 4)
        else:
            idx = A.action_index(v, labels)
        if idx < A.BLUE_RAW_ACTION:                     # (an action object naming a router comes back as a (type, host) code)
            idx = range(len(labels))[idx]               # list semantics: a negative index counts from the end, out of range raises IndexError
            idx = idx if idx < (242 if b == 4 else 82) else labels.index('Sleep')   # an explicit Sleep, not "no action" (-1)
        dur = getattr(v, 'duration', None)
        if dur is not None and not isinstance(v, (int, np.integer)) and int(dur) != A.DURATION.get(name, 1):
            if not 1 <= int(dur) <= 255:
                raise ValueError(f'{name}: duration {dur} out of range')
            idx |= int(dur) << 20                       # cc4.h cc4_step_ex: `action.duration` rides in bits 20..27
        return idx

    def _host_maps(self):
        ips = self.get_ip_map()
        t = self.topology()
        return ({str(ip): h for h, ip in ((hid, ips[host_name(hid)]) for hid in range(137) if t[27 + 2 * hid])},
                {host_name(hid): hid for hid in range(137) if t[27 + 2 * hid]},
                [f"10.0.{int(t[i])}.0/24" for i in range(9)])

    def _submit(self, actions, messages, skip_valid_action_check=False, blue_codes=None):
        """actions: {agent: action}.  A blue agent's action is a wrapper index or an action object; a red or green agent's is an action
        object (cage_challenge_4_amd.actions, or any object with the same class name and attributes -- the reference's own Action
        instances map the same way).  Agents without an entry act by the scenario's policy: on the device for the built-in classes, by
        their host-side object for custom classes / `agents=` overrides (asked here, in agent_interfaces order: blue, green, red)."""
        actions = dict(actions or {})
        self._events_fresh = False
        if self._host_agents:
            d = json_loads(self._tsjson())
            for name, obj in self._host_agents.items():
                if name in actions or (blue_codes is not None and name in self.agents_blue and blue_codes[self.agents_blue.index(name)] >= 0):
                    continue
                if name.startswith('green') and int(name.split('_')[-1]) >= d['n_green']:
                    continue
                # AgentInterface.get_action (Shared/AgentInterface.py:120-143): an inactive agent sleeps, its object is not asked
                if name.startswith('red') and not d['red'][int(name[-1])]['active']:
                    continue
                act = obj.get_action(self.get_observation(name), self.get_action_space(name))
                if act is not None:
                    actions[name] = act
        acts = np.full((1, 5), -1, np.int32) if blue_codes is None else np.asarray(blue_codes, np.int32).reshape(1, 5).copy()
        red = green = None
        maps = None
        for a, v in actions.items():
            if a in self.agents_blue:
                acts[0, self.agents_blue.index(a)] = self._blue_code(a, v)
                continue
            kind = 'red' if a.startswith('red_agent_') else ('green' if a.startswith('green_agent_') else None)
            if kind is None:
                raise ValueError(f'{a}: no such agent')
            k = int(a.split('_')[-1])
            if maps is None:
                maps = self._host_maps()
                gh = json_loads(self._tsjson())['green_hosts']
            if kind == 'red':
                if not 0 <= k < 6:
                    raise ValueError(f'{a}: no such agent')
                red = self.vec.agent_actions('red') if red is None else red
                rec, own = red[0, k], None
            else:
                if not 0 <= k < len(gh):
                    raise ValueError(f'{a}: no such agent')
                green = self.vec.agent_actions('green') if green is None else green
                rec, own = green[0, k], gh[k]
            if getattr(v, 'agent', a) != a:           # the 'agent' parameter is an ActionSpace key: another agent's name is invalid (SC:1094-1110)
                fields = (A.RED_INVALID if kind == 'red' else A.GREEN_INVALID, 0, 0, 0, 0, 0, 0.0, 0.0)
            else:
                fields = A.encode_agent_action(v, kind, maps[0], maps[1], maps[2], own_host=own, skip_valid=skip_valid_action_check)
            for f, val in zip(('type', 'host', 'arg', 'ticks', 'session', 'flags', 'rate0', 'rate1'), fields):
                rec[f] = val
        self._submitted = {a: v for a, v in actions.items() if a not in self.agents_blue}
        msg = None                                         # no messages: the engine reads zeros (the reference's EMPTY_MESSAGE)
        if messages:
            msg = np.zeros((1, 5, MESSAGE_LENGTH), np.uint8)
            for b, a in enumerate(self.agents_blue):
                m = np.asarray(messages.get(a, EMPTY_MESSAGE)).astype(bool)
                assert m.shape == (MESSAGE_LENGTH,), \
                    f'{a} attempting to send message {m} that is not in the message space MultiBinary({MESSAGE_LENGTH})'
                msg[0, b] = m
        if red is None and green is None:
            obs, rew, done, vinfo = self.vec.step(acts, msg)
        else:
            obs, rew, done, vinfo = self.vec.step_ex(acts, msg, red, green)
        raise_on_engine_error(vinfo['err'])               # ValueError past the last step (State.py:539-540); anything else is loud too
        return obs, rew, done

    def parallel_step(self, actions=None, messages=None, skip_valid_action_check=False):
        """env.py:95-123: ({agent: dict observation}, {agent: reward components of its team}, {agent: done}, {}) for the
        agents that acted plus the active ones.  Blue and red agents get their dict observation (get_observation); a green
        agent's carries 'success' and 'action'.  Every agent is listed with its team's reward and the done flag."""
        self._submit(actions, messages, skip_valid_action_check)
        st = self._state()
        rewards = self.get_rewards()
        agents = list(dict.fromkeys(list((actions or {}).keys()) + self.active_agents))
        obs = self._all_observations(st, agents)
        team = lambda a: 'Blue' if a.startswith('blue') else ('Red' if a.startswith('red') else 'Green')   # noqa: E731
        return obs, {a: dict(rewards[team(a)]) for a in agents}, {a: bool(st.raw['done']) for a in agents}, {}

    def step(self, agent=None, action=None, messages=None, skip_valid_action_check=False):
        """env.py:125-161: the single-agent step of older challenges; returns a Results-like object."""
        self._submit({} if (agent is None or action is None) else {agent: action}, messages, skip_valid_action_check)
        if agent is None:
            return Results(observation={})
        st = self._state()
        rew = self.get_rewards()['Blue' if agent.startswith('blue') else ('Red' if agent.startswith('red') else 'Green')]
        return Results(observation=self._all_observations(st, [agent])[agent],
                       done=bool(st.raw['done']), reward=round(sum(rew.values()), 1),
                       action_space=self.get_action_space(agent),
                       action=[st.last_action[agent]] if agent in st.last_action else [self._submitted.get(agent, A.Sleep())])

    def start(self, steps=None, log_file=None, verbose=False):
        """env.py:163-179 -> SimulationController.start (SC:905-950): run `steps` steps in which no action is submitted (every agent's own
        policy acts), stop early once the episode is done, return the done flag; steps=None runs until done.  (The reference's log_file
        line reads attributes of a 'Red' agent interface that CC4 scenarios do not have and raises KeyError there; not mirrored --
        log_file is ignored.  Its unconditional debugging prints are emitted only with verbose=True.)"""
        done, n = False, 0
        while steps is None or n < int(steps):
            n += 1
            self._submit({}, None)
            if verbose:
                print(n)
            done = bool(self._state().raw['done'])
            if done:
                break
        return done

    def get_agent_ids(self):
        """env.py:417-424: the keys of the controller's agent interfaces, in the scenario's order (ESG.py create_scenario: blue, green, red)."""
        n_green = int(self._state().raw['n_green'])
        return list(self.agents_blue) + [f'green_agent_{g}' for g in range(n_green)] + [f'red_agent_{r}' for r in range(6)]

    def get_message_space(self, agent):
        """env.py:449-462 -> SC:806-809: MultiBinary(message_length) for any agent name."""
        return MultiBinary(MESSAGE_LENGTH)

    def get_observation_space(self, agent):
        """env.py:284-298 -> SC:1000-1015 -> AgentInterface.get_observation_space (Shared/AgentInterface.py:199-201): the reference raises
        NotImplementedError for every agent of the scenario and ValueError for a name that is not one; so does this mirror."""
        if agent not in self.get_agent_ids():
            raise ValueError(f'Agent {agent} not in agent list {self.get_agent_ids()}')
        raise NotImplementedError

    def close(self, **kwargs):
        """env.py:426-437 ("designed for the emulator"; in the CC4 trim it reads a GUI attribute that no longer exists and raises
        AttributeError).  Here: releases the episode's device handle; calling it twice is fine."""
        vec, self.vec = getattr(self, 'vec', None), None
        if vec is not None:
            vec.close()

    def _all_observations(self, st, agents):
        """The dict observations of `agents` after the last step: blue from the end-of-turn Monitor's event log, red from the
        engine's per-agent observation keys (true_state.red_observations), green 'success' / 'action' of its own action."""
        from .true_state import blue_observations, red_observations, Ternary
        out, blue, red = {}, None, None
        for a in agents:
            if a.startswith('blue'):
                blue = blue if blue is not None else blue_observations(st)
                out[a] = blue[a]
            elif a.startswith('red'):
                red = red if red is not None else red_observations(st)
                out[a] = red[a]
            else:
                g = int(a.split('_')[-1])
                sub = getattr(self, '_submitted', {}).get(a)
                fail = (st.raw['green_fail'][g >> 5] >> (g & 31)) & 1 if 'green_fail' in st.raw else 0
                if sub is not None:
                    nm = getattr(sub, 'name', None) or type(sub).__name__
                    out[a] = {'success': Ternary('UNKNOWN' if nm == 'Sleep' else ('FALSE' if fail else 'TRUE')),
                              'action': sub if not (fail and nm not in ('GreenLocalWork', 'GreenAccessService')) else A.InvalidAction(action=sub)}
                else:                       # a device-side green agent: only whether its action failed is kept (the reward reads nothing else)
                    out[a] = {'success': Ternary('FALSE' if fail else 'UNKNOWN')}
        return out

    def edit_state(self, op, a0=0, a1=0, a2=0):
        """cc4_edit_state on this episode -- what the reference's tests do to env.environment_controller.state by hand (include/cc4.h lists the
        ops).  Hostnames and subnet names may be given instead of ids."""
        def ident(v):
            if isinstance(v, str):
                return SUBNET_NAMES.index(v) if v in SUBNET_NAMES else self._host_maps()[1][v]
            return int(v)
        return self.vec.edit_state(0, int(op), ident(a0), ident(a1), int(a2))

    def set_seed(self, seed):
        """env.py:316-325: a fresh Generator for all further randomness; the episode itself is untouched."""
        self.vec.set_seed(np.array([seed], np.uint64))

    def get_rewards(self):
        """env.py:348-357: the last step's rewards by team and component (SimulationController.py:303-311)."""
        d = self._state().raw
        ac = d['action_cost']
        brm = d['reward'] - ac
        blue = {'BlueRewardMachine': int(brm) if float(brm).is_integer() else brm}
        red, green = {'None': 0.0}, {'None': 0.0}
        if d['step'] > 0:                      # 'action_cost' appears with the first step (SimulationController.py:311)
            blue['action_cost'] = int(ac) if float(ac).is_integer() else ac
            red['action_cost'] = 0
            green['action_cost'] = 0
        return {'Blue': blue, 'Red': red, 'Green': green}

    def get_reward_breakdown(self, agent):
        """env.py:359-372 -> SimulationController.get_reward_breakdown (SC:1114-1116), which reads an attribute AgentInterface
        does not have: the reference raises AttributeError for every agent, and so does this mirror."""
        raise AttributeError("'AgentInterface' object has no attribute 'reward_calculator'")

    agents_blue = [f'blue_agent_{b}' for b in range(5)]     # (read-only: every use indexes or iterates it)

    @property
    def active_agents(self):
        """env.py:405-415 -> SimulationController.get_active_agents (SC:373-388): agents holding an active parent-less session:
        the five blue agents, every green agent, and the red agents that currently own a session without a parent."""
        d = self._state().raw
        out = list(self.agents_blue) + [f'green_agent_{g}' for g in range(d['n_green'])]
        for r, ag in enumerate(d['red']):
            if any(not (fl & 8) for _sid, _h, _pid, fl in ag['sessions']):   # RS_CHILD: session.parent is not None
                out.append(f'red_agent_{r}')
        return out

    def get_action_space(self, agent):
        """env.py:266-283 for a blue agent: the parameter dictionaries of its ActionSpace (Shared/ActionSpace.py:105-126) that
        the fixed-index wrappers are built from: the action classes, its subnets, and the known hostnames / addresses."""
        if agent.startswith('red_agent_') or agent.startswith('green_agent_'):
            return self._other_action_space(agent)
        if agent not in self.agents_blue:
            raise ValueError(f'Agent {agent} not in agent list {self.agents_blue}')
        b = self.agents_blue.index(agent)
        t = self.topology()
        cidr = self.get_cidr_map()
        ips = self.get_ip_map()
        from ipaddress import IPv4Network, IPv4Address
        mine = {h for h in ips if any(h.startswith(sn) for sn in BLUE_SUBNETS[b])}
        return {'action': {c: True for c in A.BLUE_ACTIONS},
                'allowed_subnets': list(BLUE_SUBNETS[b]),
                'subnet': {IPv4Network(cidr[sn]): sn in BLUE_SUBNETS[b] for sn in SUBNET_NAMES},
                'ip_address': {IPv4Address(ip): h in mine for h, ip in ips.items()},
                'session': {0: True},
                'username': {'root': True, 'user': True}, 'password': {},
                'agent': {agent: True},
                'hostname': {h: h in mine for h in ips}}

    def _other_action_space(self, agent):
        """ActionSpace.get_action_space (Shared/ActionSpace.py:105-126) of a red or green agent from the engine's ActionSpace bits
        (RedAgent.as_ip / as_hn / as_subnet / known sessions): every address / hostname / subnet of the episode, True where known."""
        from ipaddress import IPv4Network, IPv4Address
        d = self._state().raw
        cidr, ips = self.get_cidr_map(), self.get_ip_map()
        k = int(agent.split('_')[-1])
        bit = lambda words, h: bool((words[h >> 5] >> (h & 31)) & 1)      # noqa: E731
        hid = {host_name(h['h']): h['h'] for h in d['hosts']}
        if agent.startswith('red'):
            ag = d['red'][k]
            allowed = [['contractor_network_subnet'], ['restricted_zone_a_subnet'], ['operational_zone_a_subnet'], ['restricted_zone_b_subnet'],
                       ['operational_zone_b_subnet'], ['public_access_zone_subnet', 'admin_network_subnet', 'office_network_subnet']][k]   # ESG.py:769-776
            return {'action': {c: True for c in A.RED_ACTIONS}, 'allowed_subnets': allowed,
                    'subnet': {IPv4Network(cidr[sn]): bool((ag['as_subnet'] >> i) & 1) for i, sn in enumerate(SUBNET_NAMES)},
                    'ip_address': {IPv4Address(ips[n]): bit(ag['as_ip'], h) for n, h in hid.items()},
                    'session': {sid: True for sid in ag['known_sessions']},
                    'username': {}, 'password': {}, 'process': {}, 'port': {}, 'target_session': {sid: True for sid in ag['known_sessions']},
                    'agent': {agent: True}, 'hostname': {n: bit(ag['as_hostname'], h) for n, h in hid.items()}}
        gh = d['green_hosts'][k]
        sn = gh // 17
        policy = getattr(self.scenario_generator, 'green_policy', 0)
        from .true_state import green_allowed_subnets
        return {'action': {c: True for c in (A.GREEN_ACTIONS if policy != 1 else (A.Sleep,))},
                'allowed_subnets': green_allowed_subnets(d['phase'], sn),
                'subnet': {IPv4Network(cidr[SUBNET_NAMES[sn]]): True},
                'ip_address': {IPv4Address(ips[n]): True for n, h in hid.items() if h // 17 == sn},
                'session': {0: True}, 'username': {}, 'password': {}, 'process': {}, 'port': {}, 'target_session': {},
                'agent': {agent: True}, 'hostname': {n: True for n, h in hid.items() if h // 17 == sn}}

    def get_agent_state(self, agent_name):
        """env.py:202-216: the true state restricted to what the scenario's INFO_DICT lists for the agent ('True' = all)."""
        from .true_state import decode
        st = decode(self._tsjson())
        if agent_name == 'True' or agent_name not in self.agents_blue:
            return st.as_dict()
        b = self.agents_blue.index(agent_name)
        return st.as_dict({h: None for h in st.hosts if any(h.startswith(sn) for sn in BLUE_SUBNETS[b])})

    def get_attr(self, attribute):
        return getattr(self, attribute) if hasattr(self, attribute) else None

    def topology(self):
        return self.vec.topology(0)

    def get_cidr_map(self):
        t = self.topology()
        return {SUBNET_NAMES[i]: f"10.0.{int(t[i])}.0/24" for i in range(9)}

    def get_true_state(self, info=None):
        """env.py:297-309 -> State.get_true_state: {hostname: {'Interface', 'Processes', 'Sessions', 'Services',
        'System info', ...}, 'success': True}, decoded from the packed episode state (true_state.py).  `info`
        (hostname -> wanted fields) only selects hosts."""
        from .true_state import decode
        return decode(self._tsjson()).as_dict(info)

    def get_observation(self, agent):
        """env.py:270-283: the dict observation of a blue agent after the last step -- 'success', 'action' and, per host of
        its zone with events, 'Interface' / 'Processes' (connections with addresses and ports, pids) / 'System info', as the
        end-of-turn Monitor reports them (true_state.blue_observations; exact against the reference for every blue action,
        incl. the process entry of a resolved DeployDecoy and the file list of a resolved Analyse: tests/test_blue_obs.py);
        a red agent's comes from the engine's per-agent observation keys, a green agent's carries 'success' / 'action'."""
        from .true_state import decode
        return self._all_observations(decode(self._tsjson()), [agent])[agent]

    def get_last_action(self, agent):
        """env.py:300-314: the actions of `agent` (blue_agent_b / red_agent_r) that resolved in the last step -- a list, as the
        reference returns (one entry here) -- as objects whose str() equals the reference action's ('Restore <hostname>',
        'ExploitRemoteService <ip>', 'Sleep', ...) and whose .name is the action class name."""
        from .true_state import decode
        return [decode(self._tsjson()).last_action[agent]]

    def get_ip_map(self):
        t = self.topology()
        out = {}
        for h in range(137):
            if t[27 + 2 * h]:
                out[host_name(h)] = f"10.0.{int(t[h // 17])}.{int(t[28 + 2 * h])}"
        return out


class Results:
    """Shared/Results.py:21-60, the fields CybORG.step fills."""
    def __init__(self, observation=None, done=None, reward=None, action_space=None, action=None, info=None, error=None):
        self.observation, self.done, self.reward = observation, done, reward
        self.action_space, self.action, self.info, self.error = action_space, action, info, error


def build_action_labels(cidr, masks, pad_spaces=False, max_size=242):
    """The fixed action list of every blue agent as labels + mask (BlueFixedActionWrapper.py:233-309): Analyse / Monitor /
    Remove / Restore / Sleep / AllowTrafficZone / BlockTrafficZone / DeployDecoy over the agent's sorted hosts and subnets;
    the mask comes from the engine (which host slots exist in this episode)."""
    out = {}
    for b in range(5):
        a = f'blue_agent_{b}'
        mask = [bool(v) for v in masks[b][0]]
        hosts = sorted(f'{sn}_{kind}_host_{i}' for sn in BLUE_SUBNETS[b]
                       for kind, n in (('user', MAX_USER_HOSTS), ('server', MAX_SERVER_HOSTS)) for i in range(n))
        subnets = sorted(BLUE_SUBNETS[b])
        labels = []

        def host_block(cmd):
            base = len(labels)
            for i, h in enumerate(hosts):
                labels.append(f"{cmd} {h}" if mask[base + i] else f"[Invalid] {cmd} {h}")

        def zone_block(cmd):
            for dst in subnets:
                for src in sorted(SUBNET_NAMES):
                    if src == dst:
                        continue
                    labels.append(f"{cmd} {dst} ({cidr[dst]}) <- {src} ({cidr[src]})")
        host_block('Analyse'); labels.append('Monitor'); host_block('Remove'); host_block('Restore'); labels.append('Sleep')
        zone_block('AllowTrafficZone'); zone_block('BlockTrafficZone'); host_block('DeployDecoy')
        assert len(labels) == len(mask)
        if pad_spaces and len(labels) < max_size:
            pad = max_size - len(labels)
            labels += ['[Padding] Sleep'] * pad
            mask += [False] * pad
        out[a] = {'actions': list(range(len(labels))), 'labels': labels, 'mask': mask}
    return out


def host_name(h):
    if h == 136:
        return 'root_internet_host_0'
    s, slot = divmod(h, 17)
    if slot == 0:
        return f'{SUBNET_NAMES[s]}_router'
    if slot <= 10:
        return f'{SUBNET_NAMES[s]}_user_host_{slot - 1}'
    return f'{SUBNET_NAMES[s]}_server_host_{slot - 11}'


class BlueFixedActionWrapper:
    """BlueFixedActionWrapper.py:26-379 -- fixed int action index per blue agent, masks and labels."""
    def __init__(self, env: CybORG, pad_spaces: bool = False, *args, **kwargs):
        self.env = env
        self.agents = self.possible_agents = list(env.agents)
        self._pad_spaces = pad_spaces
        self._max_act_space_size = 242
        self._agent_metadata = {}
        for b, a in enumerate(self.possible_agents):
            hosts = set()
            for sn in BLUE_SUBNETS[b]:
                hosts.add(f'{sn}_router')
                hosts.update(f'{sn}_user_host_{i}' for i in range(MAX_USER_HOSTS))
                hosts.update(f'{sn}_server_host_{i}' for i in range(MAX_SERVER_HOSTS))
            self._agent_metadata[a] = {'hosts': sorted(hosts), 'subnets': sorted(BLUE_SUBNETS[b])}
        if hasattr(env, '_event_log_on_demand'):
            env._event_log_on_demand()
        self._refresh_action_space()

    # -- action space bookkeeping (mask comes from the engine, labels are rebuilt from the topology)
    def _refresh_action_space(self):
        self._action_space = build_action_labels(self.env.get_cidr_map(), split_mask(self.env.vec.action_mask),
                                                 self._pad_spaces, self._max_act_space_size)

    def reset(self, *args, **kwargs):
        self.env.reset(*args, **kwargs)
        self.agents = self.possible_agents
        self._refresh_action_space()
        obs = split_obs(self.env.vec._obs)
        observations = {a: obs[b][0].astype(np.int64) for b, a in enumerate(self.agents)}
        info = {a: {'action_mask': self._action_space[a]['mask']} for a in self.agents}
        return observations, info

    def step(self, actions=None, messages=None, **kwargs):
        action_dict = {} if actions is None else actions
        acts = [-1] * 5
        others = {}
        for a, v in action_dict.items():
            if a not in self.possible_agents:
                # an action object for a red or green agent: the reference's wrapper forwards it to parallel_step as it is
                # (BlueFixedActionWrapper.py:142-148 maps ints and passes Action objects through)
                others[a] = v
                continue
            # an index into the agent's action list (Python list semantics: negative counts from the end, out of range raises
            # IndexError), or an action object, which the reference forwards as it is (BlueFixedActionWrapper.py:142-148).  A padded
            # slot ('[Padding] Sleep') is an explicit Sleep() submitted by the agent (:320-332) -- never -1, which means "no action
            # submitted" and hands the agent to the scenario's blue policy
            acts[self.possible_agents.index(a)] = self.env._blue_code(a, v, self._action_space[a]['labels'])
        # State.check_next_phase_on_update_step (State.py:539-540) raises ValueError past the last step; any other engine flag
        # (a container bound, a path the reference would crash on) raises CC4EngineError: never a silently different result
        obs, rew, done = self.env._submit(others, messages, kwargs.get('skip_valid_action_check', False), blue_codes=acts)
        d = bool(done[0])
        row = obs[0].astype(np.int64)                       # one conversion; the agents' vectors are disjoint slices of it
        agents = self.possible_agents
        observations = {a: row[o:o + n] for a, o, n in zip(agents, L.OBS_OFF, L.OBS_LEN)}
        r = float(rew[0])
        rewards = dict.fromkeys(agents, r)
        terminated = dict.fromkeys(agents, d)
        truncated = dict.fromkeys(agents, d)
        info = {a: {'action_mask': self._action_space[a]['mask']} for a in agents}
        self.agents = [] if d else agents
        return observations, rewards, terminated, truncated, info

    def get_action_space(self, agent):
        return self._action_space[agent]

    def hosts(self, agent_name):
        return self._agent_metadata[agent_name]['hosts']

    def subnets(self, agent_name):
        return self._agent_metadata[agent_name]['subnets']

    def action_mask(self, agent_name):
        return self._action_space[agent_name]['mask']

    def action_labels(self, agent_name):
        return self._action_space[agent_name]['labels']

    def actions(self, agent_name):
        return self._action_space[agent_name]['actions']

    @property
    def is_padded(self):
        return self._pad_spaces

    def action_space(self, agent_name):
        if self._pad_spaces:
            return Discrete(self._max_act_space_size)
        return Discrete(len(self._action_space[agent_name]['actions']))

    def action_spaces(self):
        return {a: self.action_space(a) for a in self.agents}

    def get_message_space(self, agent):
        return MultiBinary(MESSAGE_LENGTH)

    # -- the BaseWrapper surface the reference's wrappers inherit (Agents/Wrappers/BaseWrapper.py:32-53)
    def get_observation(self, agent):
        """BaseWrapper.py:35-36: the agent's observation after the last step, in this wrapper's form."""
        return self.env.get_observation(agent)

    def get_last_action(self, agent):
        return self.env.get_last_action(agent)

    def set_seed(self, seed):
        self.env.set_seed(seed)

    @property
    def active_agents(self):
        return self.env.active_agents

    def get_attr(self, attribute):
        if hasattr(self, attribute):
            return getattr(self, attribute)
        return getattr(self.env, attribute, None)

    @property
    def unwrapped(self):
        return self.env

    def close(self):
        self.env.vec.close()


class BlueFlatWrapper(BlueFixedActionWrapper):
    """BlueFlatWrapper.py:31-322 -- flat int64 observation vectors (92 / 210 values)."""
    def __init__(self, env, *args, **kwargs):
        super().__init__(env, *args, **kwargs)
        middle = 9 * [2] + 9 * [2] + 16 * [2] + 16 * [2] + 9 * [2]
        self._short_obs_space = MultiDiscrete([3] + middle + 32 * [2])
        self._long_obs_space = MultiDiscrete([3] + 3 * middle + 32 * [2])

    def _pad(self, observations):
        if not self.is_padded:
            return observations
        return {a: np.pad(o, (0, 210 - o.shape[0])) for a, o in observations.items()}

    def reset(self, *args, **kwargs):
        observations, info = super().reset(*args, **kwargs)
        return self._pad(observations), info

    def step(self, actions=None, messages=None, **kwargs):
        observations, rewards, terminated, truncated, info = super().step(actions=actions, messages=messages, **kwargs)
        return self._pad(observations), rewards, terminated, truncated, info

    def get_observation(self, agent):
        """BaseWrapper.get_observation through BlueFlatWrapper.observation_change (BlueFlatWrapper.py:172-256): a blue agent's flat
        vector after the last step (what the reference's own wrapper tests read between steps: Tests/test_cc4/test_BlueEnterpriseWrapper.py:196-199,338)."""
        if agent in self.possible_agents:
            o = split_obs(self.env.vec._obs)[self.possible_agents.index(agent)][0].astype(np.int64)
            return np.pad(o, (0, 210 - o.shape[0])) if self.is_padded else o
        return self.env.get_observation(agent)

    def observation_space(self, agent_name):
        return self._long_obs_space if (self.is_padded or agent_name == 'blue_agent_4') else self._short_obs_space

    def observation_spaces(self):
        return {a: self.observation_space(a) for a in self.possible_agents}


class BlueEnterpriseWrapper(BlueFlatWrapper):
    """BlueEnterpriseWrapper.py:25-122 -- accepts {"actions": ..., "messages": ...}."""
    def step(self, actions=None, messages=None):
        action_dict = actions if actions is not None else {}
        if 'actions' in action_dict:
            messages = action_dict.get('messages', messages)
            return super().step(action_dict['actions'], messages=messages)
        return super().step(action_dict, messages=messages)

    def reset(self, agent=None, seed=None, *args, **kwargs):
        return super().reset(agent=agent, seed=seed)

    @property
    def long_observation_space(self):
        return self._long_obs_space

    @property
    def short_observation_space(self):
        return self._short_obs_space


class EnterpriseMAE(BlueEnterpriseWrapper):
    """EnterpriseMAE.py:10-72 -- RLlib MultiAgentEnv flavour: adds the "__all__" keys."""
    def step(self, action_dict=None, messages=None):
        obs, rew, terminated, truncated, info = BlueFlatWrapper.step(self, actions=action_dict, messages=messages)
        done = bool(truncated[self.possible_agents[0]])
        terminated['__all__'] = False
        truncated['__all__'] = done
        return obs, rew, terminated, truncated, info
