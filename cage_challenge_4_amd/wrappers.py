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
import os
import ctypes
import numpy as np
from . import _lib as L
from .vec_env import CC4VecEnv, RNG_PCG64, split_obs, split_mask
from .spaces import Discrete, MultiDiscrete, MultiBinary

SUBNET_NAMES = ['restricted_zone_a_subnet', 'operational_zone_a_subnet', 'restricted_zone_b_subnet',
                'operational_zone_b_subnet', 'contractor_network_subnet', 'public_access_zone_subnet',
                'admin_network_subnet', 'office_network_subnet', 'internet_subnet']   # SUBNET enum order (ESG.py:40-51)
BLUE_SUBNETS = [['restricted_zone_a_subnet'], ['operational_zone_a_subnet'], ['restricted_zone_b_subnet'],
                ['operational_zone_b_subnet'],
                ['public_access_zone_subnet', 'admin_network_subnet', 'office_network_subnet']]  # ESG.py:643-649
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

    def __init__(self, blue_agent_class=None, red_agent_class=None, green_agent_class=None, steps: int = 100):
        if blue_agent_class not in (None, SleepAgent) and getattr(blue_agent_class, '__name__', '') != 'SleepAgent':
            raise NotImplementedError("blue default policy: only SleepAgent (blue actions are submitted through step())")
        red = {'FiniteStateRedAgent': 0, 'SleepAgent': 1, 'DiscoveryFSRed': 2, 'RandomSelectRedAgent': 3}
        green = {'EnterpriseGreenAgent': 0, 'SleepAgent': 1}
        # None -> SleepAgent, as the reference's generators do (ESG.py:745-748, :815-817)
        rn = 'SleepAgent' if red_agent_class is None else getattr(red_agent_class, '__name__', None)
        gn = 'SleepAgent' if green_agent_class is None else getattr(green_agent_class, '__name__', None)
        if rn not in red or gn not in green:
            raise NotImplementedError(f"built-in policies of the HIP engine: red {list(red)}, green {list(green)} "
                                      f"(got red={red_agent_class}, green={green_agent_class}); "
                                      "other policies are SURVEY 8(f) 'next' rows")
        self.red_policy, self.green_policy = red[rn], green[gn]
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
        if agents and not isinstance(agents, str):   # evaluation.py:69 passes the string "sim" here; it selects nothing
            raise NotImplementedError("per-agent policy overrides are not part of the accelerated path")
        self.scenario_generator = scenario_generator
        if seed is None:
            seed = int.from_bytes(os.urandom(8), 'little') >> 1
        if not isinstance(seed, (int, np.integer)):
            raise NotImplementedError("custom Generator objects cannot be injected into the device RNG; pass an int seed")
        # vec_factory: anything with CC4VecEnv's call shape (tests inject the CPU oracle; the default is the HIP engine)
        self.vec = (vec_factory or CC4VecEnv)(1, steps=scenario_generator.steps, rng_mode=rng_mode, device_id=device_id,
                                              red_policy=scenario_generator.red_policy,
                                              green_policy=scenario_generator.green_policy)
        self.vec.enable_event_log(True)                        # single episode: keep the per-step event detail for get_observation
        self.vec.reset(seeds=np.array([seed], np.uint64))      # SimulationController.__init__ creates a scenario
        self.agents = [f'blue_agent_{b}' for b in range(5)]

    def reset(self, agent=None, seed=None):
        """env.py:218-243: seed=None keeps the running stream; an int seed starts a fresh Generator."""
        self.vec.reset(seeds=None if seed is None else np.array([seed], np.uint64))

    @property
    def unwrapped(self):
        return self

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
        return decode(self.vec.true_state_json(0)).as_dict(info)

    def get_observation(self, agent):
        """env.py:270-283: the dict observation of a blue agent after the last step -- 'success', 'action' and, per host of
        its zone with events, 'Interface' / 'Processes' (connections with addresses and ports, pids) / 'System info', as the
        end-of-turn Monitor reports them (true_state.blue_observations; exact against the reference for Sleep / Monitor /
        Remove / Restore / Block / Allow steps)."""
        from .true_state import decode, blue_observations
        return blue_observations(decode(self.vec.true_state_json(0)))[agent]

    def get_last_action(self, agent):
        """env.py:300-314: the action of `agent` (blue_agent_b / red_agent_r) that resolved in the last step, as an object
        whose str() equals the reference action's ('Restore <hostname>', 'ExploitRemoteService <ip>', 'Sleep', ...)."""
        from .true_state import decode
        return decode(self.vec.true_state_json(0)).last_action[agent]

    def get_ip_map(self):
        t = self.topology()
        out = {}
        for h in range(137):
            if t[27 + 2 * h]:
                out[host_name(h)] = f"10.0.{int(t[h // 17])}.{int(t[28 + 2 * h])}"
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
        self._refresh_action_space()

    # -- action space bookkeeping (mask comes from the engine, labels are rebuilt from the topology)
    def _refresh_action_space(self):
        masks = split_mask(self.env.vec.action_mask)
        cidr = self.env.get_cidr_map()
        self._action_space = {}
        for b, a in enumerate(self.possible_agents):
            mask = [bool(v) for v in masks[b][0]]
            hosts = [h for h in self._agent_metadata[a]['hosts'] if 'router' not in h]
            subnets = self._agent_metadata[a]['subnets']
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
            if self._pad_spaces and len(labels) < self._max_act_space_size:
                pad = self._max_act_space_size - len(labels)
                labels += ['[Padding] Sleep'] * pad
                mask += [False] * pad
            self._action_space[a] = {'actions': list(range(len(labels))), 'labels': labels, 'mask': mask}

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
        acts = np.full((1, 5), -1, np.int32)
        for a, v in action_dict.items():
            if not isinstance(v, (int, np.integer)):
                raise NotImplementedError("the accelerated path takes wrapper action indices, not Action objects")
            b = self.possible_agents.index(a)
            n = 242 if b == 4 else 82
            acts[0, b] = int(v) if 0 <= int(v) < n else -1      # padded slots are Sleep (BlueFixedActionWrapper.py:320-332)
            if not 0 <= int(v) < len(self._action_space[a]['labels']):
                raise IndexError('list index out of range')       # same failure as indexing the reference's action list
        messages = {} if messages is None else messages
        msg = np.zeros((1, 5, MESSAGE_LENGTH), np.uint8)
        for b, a in enumerate(self.possible_agents):
            m = np.asarray(messages.get(a, EMPTY_MESSAGE)).astype(bool)
            assert m.shape == (MESSAGE_LENGTH,), \
                f'{a} attempting to send message {m} that is not in the message space MultiBinary({MESSAGE_LENGTH})'
            msg[0, b] = m
        obs, rew, done, vinfo = self.env.vec.step(acts, msg)
        if int(vinfo['err'][0]) & (1 << 7):   # State.check_next_phase_on_update_step (State.py:539-540)
            raise ValueError("Step number exceeds last mission phase step maximum. "
                             "Use step parameter in EnterpriseScenarioGenerator.")
        d = bool(done[0])
        ob = split_obs(obs)
        observations = {a: ob[b][0].astype(np.int64) for b, a in enumerate(self.possible_agents)}
        rewards = {a: float(rew[0]) for a in self.possible_agents}
        terminated = {a: d for a in self.possible_agents}
        truncated = {a: d for a in self.possible_agents}
        info = {a: {'action_mask': self._action_space[a]['mask']} for a in self.possible_agents}
        self.agents = [a for a in self.possible_agents if not d]
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
