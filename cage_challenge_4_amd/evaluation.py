"""Batched counterpart of the reference's evaluation loop (CybORG/Evaluation/evaluation.py:56-196, README.md:231).

Protocol restated: EnterpriseScenarioGenerator(SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent, steps=500);
`max_eps` episodes; per step every blue agent in `submission.AGENTS` picks an action from its flat observation; the
episode score is the sum over steps of mean_a(reward), where the step on which `done` is raised is NOT counted
(evaluation.py:108-110 breaks before appending); result = mean / stdev over episodes; files summary.txt, scores.txt,
summary.json.

Two modes:
  * mode='sequential' -- one episode at a time on ONE stream, exactly like the reference (CybORG(seed); reset() per
    episode continues the stream): the per-episode scores equal the reference's under the same seed and agents.
  * mode='batched'    -- all episodes at once as one CC4VecEnv batch (episode i seeded seed+i): statistically
    equivalent, ~max_eps times fewer kernel launches.  Agents may expose `get_actions(obs[N,L], mask[N,A]) -> int[N]`
    for vectorised inference; otherwise `get_action(obs, action_space)` is called per episode.
"""
import json
import os
from datetime import datetime
from statistics import mean, stdev

import numpy as np

from . import _lib as L
from .vec_env import CC4VecEnv, RNG_PCG64, split_obs, split_mask
from .spaces import Discrete
from .wrappers import (CybORG, EnterpriseScenarioGenerator, SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent,
                       BlueFlatWrapper)

EPISODE_LENGTH = 500
CYBORG_VERSION = '4.0'


def _headers(submission):
    version_header = f"CybORG v{CYBORG_VERSION}, Scenario4"
    author_header = f"Author: {submission.NAME}, Team: {submission.TEAM}, Technique: {submission.TECHNIQUE}"
    return version_header, author_header


def _write(submission, log_path, total_reward, seed, max_eps, start, end, episode_length, actions_log=None, obs_log=None):
    if not log_path.endswith('/'):
        log_path += '/'
    os.makedirs(log_path, exist_ok=True)
    version_header, author_header = _headers(submission)
    reward_mean = mean(total_reward)
    reward_stdev = stdev(total_reward) if len(total_reward) > 1 else 0.0
    reward_string = f"Average reward is: {reward_mean} with a standard deviation of {reward_stdev}"
    with open(log_path + 'summary.txt', 'w') as f:
        f.write(version_header + '\n' + author_header + '\n' + reward_string + '\n' + f"Using agents {submission.AGENTS}")
    with open(log_path + 'summary.json', 'w') as f:
        json.dump({'submission': {'author': submission.NAME, 'team': submission.TEAM, 'technique': submission.TECHNIQUE},
                   'parameters': {'seed': seed, 'episode_length': episode_length, 'max_episodes': max_eps},
                   'time': {'start': str(start), 'end': str(end), 'elapsed': str(end - start)},
                   'reward': {'mean': reward_mean, 'stdev': reward_stdev},
                   'agents': {a: str(submission.AGENTS[a]) for a in submission.AGENTS}}, f)
    with open(log_path + 'scores.txt', 'w') as f:
        f.write(f"reward_mean: {reward_mean}\n")
        f.write(f"reward_stdev: {reward_stdev}\n")
    if actions_log is not None:      # evaluation.py:158-176: the per-episode action and observation logs
        with open(log_path + 'full.txt', 'w') as f:
            f.write(version_header + '\n' + author_header + '\n' + reward_string + '\n')
            for act, obs, sum_rew in zip(actions_log, obs_log, total_reward):
                f.write(f"actions: {act},\n observations: {obs},\n total reward: {sum_rew}\n")
        with open(log_path + 'actions.txt', 'w') as f:
            f.write(version_header + '\n' + author_header + '\n' + reward_string + '\n')
            for act in zip(actions_log):
                f.write(f"actions: {act}")


def run_evaluation(submission, log_path=None, max_eps=100, write_to_file=True, seed=None, mode='batched',
                   episode_length=EPISODE_LENGTH, vec_factory=None, device_id=0):
    """Returns the list of per-episode scores (the reference prints mean/stdev and writes the files)."""
    start = datetime.now()
    if mode == 'sequential':
        sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=EnterpriseGreenAgent,
                                         red_agent_class=FiniteStateRedAgent, steps=episode_length)
        cyborg = CybORG(sg, 'sim', seed=seed, vec_factory=vec_factory, device_id=device_id)
        env = submission.wrap(cyborg)
        total_reward = []
        actions_log, obs_log = ([], []) if (write_to_file and log_path) else (None, None)
        for _ in range(max_eps):
            observations, _info = env.reset()
            r, a_log, o_log = [], [], []
            for _j in range(episode_length):
                actions = {name: agent.get_action(observations[name], env.action_space(name))
                           for name, agent in submission.AGENTS.items() if name in env.agents}
                observations, rew, term, trunc, info = env.step(actions)
                done = {a: term.get(a, False) or trunc.get(a, False) for a in env.agents}
                if all(done.values()):
                    break
                r.append(mean(rew.values()))
                if actions_log is not None:     # evaluation.py:111-124: what resolved this step (get_last_action) and what the agents saw
                    a_log.append({name: cyborg.get_last_action(name) for name in env.agents})
                    o_log.append({name: observations[name] for name in observations.keys()})
            total_reward.append(sum(r))
            if actions_log is not None:
                actions_log.append(a_log)
                obs_log.append(o_log)
    elif mode == 'batched':
        if seed is None:
            seed = int.from_bytes(os.urandom(8), 'little') >> 1
        n = max_eps
        vec = (vec_factory or CC4VecEnv)(n, steps=episode_length, rng_mode=RNG_PCG64, device_id=device_id)
        obs = vec.reset(seeds=np.uint64(seed) + np.arange(n, dtype=np.uint64))
        masks = split_mask(vec.action_mask)
        names = [f'blue_agent_{b}' for b in range(5)]
        spaces = [Discrete(82), Discrete(82), Discrete(82), Discrete(82), Discrete(242)]
        score = np.zeros(n, np.float64)
        alive = np.ones(n, bool)
        # batched mode logs what was SUBMITTED per episode and step (labels of the fixed action list) and the observations;
        # the sequential mode logs what resolved (get_last_action), as the reference does
        actions_log, obs_log = ([[] for _ in range(n)], [[] for _ in range(n)]) if (write_to_file and log_path) else (None, None)
        for _j in range(episode_length):
            parts = split_obs(obs)
            acts = np.full((n, 5), -1, np.int32)
            for b, name in enumerate(names):
                agent = submission.AGENTS.get(name)
                if agent is None:
                    continue
                if hasattr(agent, 'get_actions'):
                    acts[:, b] = np.asarray(agent.get_actions(parts[b].astype(np.int64), masks[b]), np.int32)
                else:
                    for i in range(n):
                        if alive[i]:
                            acts[i, b] = int(agent.get_action(parts[b][i].astype(np.int64), spaces[b]))
            obs, rew, done, _info = vec.step(acts)       # raises on any engine error flag (CC4VecEnv strict mode)
            alive &= ~done
            score += np.where(alive, rew, 0.0)           # the step that raises done is not counted (evaluation.py:108-110)
            if actions_log is not None:
                po = split_obs(obs)
                for i in np.nonzero(alive)[0]:
                    actions_log[i].append({name: int(acts[i, b]) for b, name in enumerate(names) if name in submission.AGENTS})
                    obs_log[i].append({name: po[b][i].astype(np.int64) for b, name in enumerate(names)})
            if not alive.any():
                break
        total_reward = [float(x) for x in score]
        if hasattr(vec, 'close'):
            vec.close()
    else:
        raise ValueError("mode must be 'sequential' or 'batched'")
    end = datetime.now()
    reward_mean = mean(total_reward)
    reward_stdev = stdev(total_reward) if len(total_reward) > 1 else 0.0
    print(f"Average reward is: {reward_mean} with a standard deviation of {reward_stdev}")
    if write_to_file and log_path:
        _write(submission, log_path, total_reward, seed, max_eps, start, end, episode_length, actions_log, obs_log)
    return total_reward
