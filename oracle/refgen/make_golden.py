"""Generate the golden trajectories under tests/golden/ from the REAL reference (CybORG v4 at /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  Each fixture is data only:
seed, init mode, per-step blue action indices, and the reference's outputs: flat observations (5 agents
concatenated, 578 values), team reward, done flag, action mask, and the numpy PCG64 stream position after
every step (128-bit state + has_uint32 + uinteger), so that a restatement can be pinned bit-for-bit including
its draw order.  numpy version is recorded (reference pins 1.26.4; this container has 2.x -- stream algorithms
unchanged between them as far as the documented Generator API guarantees).

usage: python make_golden.py            # writes tests/golden/traj_*.npz
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Agents import SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent, DiscoveryFSRed, RandomSelectRedAgent, cc4BlueRandomAgent
RED = {'fsm': (FiniteStateRedAgent, 0), 'sleep': (SleepAgent, 1), 'discovery': (DiscoveryFSRed, 2), 'random': (RandomSelectRedAgent, 3)}
GREEN = {'enterprise': (EnterpriseGreenAgent, 0), 'sleep': (SleepAgent, 1)}
from CybORG.Agents.Wrappers import BlueFlatWrapper
from blue_policies import BluePolicy, KINDS

OUT = os.path.join(os.path.dirname(__file__), '..', '..', 'tests', 'golden')


def rng_words(env):
    st = env.environment_controller.np_random.bit_generator.state
    v = st['state']['state']
    return [v >> 64, v & ((1 << 64) - 1), int(st['has_uint32']), int(st['uinteger'])]


def flat(obs):
    return np.concatenate([obs[f'blue_agent_{b}'] for b in range(5)]).astype(np.uint8)


def record(seed, steps, blue, init, nsteps=None, msgs=False, red='fsm', green='enterprise'):
    # blue 'builtin': blue_agent_class=cc4BlueRandomAgent and no action submitted -- the scenario's own agent objects act
    sg = EnterpriseScenarioGenerator(blue_agent_class=cc4BlueRandomAgent if blue == 'builtin' else SleepAgent,
                                     green_agent_class=GREEN[green][0], red_agent_class=RED[red][0], steps=steps)
    env = CybORG(sg, seed=seed)
    w = BlueFlatWrapper(env)
    if init == 'ctor':      # CybORG(seed=s); wrapper.reset()  -> second scenario drawn from the running stream
        obs, info = w.reset()
        reset_seed = -1
    else:                   # wrapper.reset(seed=s+1)          -> fresh Generator
        obs, info = w.reset(seed=seed + 1)
        reset_seed = seed + 1
    arng = np.random.default_rng(seed ^ 0xB10E)
    bpol = BluePolicy(blue, {f'blue_agent_{b}': w.action_labels(f'blue_agent_{b}') for b in range(5)}, seed) if blue in KINDS else None
    nsteps = nsteps or steps
    A = np.full((nsteps, 5), -1, np.int16)
    O = np.zeros((nsteps + 1, 578), np.uint8)
    R = np.zeros(nsteps, np.float32)
    D = np.zeros(nsteps, np.uint8)
    G = np.zeros((nsteps + 1, 4), np.uint64)
    M = np.zeros((nsteps, 5, 8), np.uint8)
    mask = np.concatenate([np.array(info[f'blue_agent_{b}']['action_mask'], np.uint8) for b in range(5)])
    O[0] = flat(obs)
    G[0] = rng_words(env)
    for t in range(nsteps):
        acts = {}
        if blue == 'random':
            A[t] = [arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(242)]
            acts = {f'blue_agent_{b}': int(A[t, b]) for b in range(5)}
        elif bpol is not None:          # structured blue policy (blue_policies.py): one kind of action per agent
            A[t] = bpol.act(t)
            acts = {f'blue_agent_{b}': int(A[t, b]) for b in range(5)}
        messages = None
        if msgs:
            M[t] = arng.integers(0, 2, size=(5, 8))
            messages = {f'blue_agent_{b}': M[t, b].astype(bool) for b in range(5)}
        obs, rew, term, trunc, info = w.step(acts, messages=messages)
        O[t + 1] = flat(obs)
        R[t] = rew['blue_agent_0']
        assert len(set(rew.values())) == 1
        D[t] = term['blue_agent_0']
        G[t + 1] = rng_words(env)
    pol = ('' if red == 'fsm' else f'_red{red}') + ('' if green == 'enterprise' else f'_green{green}')
    name = f"traj_seed{seed}_{blue}_{init}_{steps}{'_msg' if msgs else ''}{pol}.npz"
    np.savez_compressed(os.path.join(OUT, name), seed=np.int64(seed), reset_seed=np.int64(reset_seed), steps=np.int32(steps),
                        actions=A, obs_bits=np.packbits(O[:, 1:] if False else (O > 0).astype(np.uint8), axis=1), phase=O[:, 0].copy(),
                        phase_cols=np.array([0, 92, 184, 276, 368], np.int32), obs_phase_vals=O[:, [0, 92, 184, 276, 368]].copy(),
                        reward=R, done=D, rng=G, mask=mask, messages=M if msgs else np.zeros(0, np.uint8),
                        red_policy=np.int32(RED[red][1]), green_policy=np.int32(GREEN[green][1]), blue_policy=np.int32(blue == 'builtin'),
                        numpy_version=np.bytes_(np.__version__), n_hosts=np.int32(len(env.environment_controller.state.hosts)))
    print(name, 'sum reward', float(R.sum()), 'hosts', len(env.environment_controller.state.hosts), flush=True)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'policies':   # built-in policy variants
        record(41, 500, 'random', 'ctor', red='discovery')
        record(42, 300, 'random', 'reset', red='sleep')
        record(43, 200, 'random', 'ctor', red='discovery', green='sleep')
        record(44, 500, 'random', 'ctor', red='random')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'builtin':   # cc4BlueRandomAgent as the scenario's blue agent
        record(123, 500, 'builtin', 'ctor')
        record(87, 500, 'builtin', 'ctor', red='random')     # a seed of test_heuristic_agents.py::test_sessions_issue_with_blue
        record(45, 300, 'builtin', 'reset', red='discovery')
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'long':   # 1000-step episodes (r03: the cold containers are sized from the episode length), as in
        record(777, 1000, 'decoy_one', 'ctor')           # CybORG/Tests/test_cc4/test_heuristic_agents.py:10-84; 500 decoys stacked on one host
        record(87, 1000, 'random', 'ctor', red='random')  # a seed of that test with RandomSelectRedAgent
        record(4000, 1000, 'builtin', 'reset', red='random')   # the test's own set-up: cc4BlueRandomAgent blue, RandomSelectRedAgent red
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == 'structured':   # structured blue policies (r02: the process lists are unbounded)
        record(321, 500, 'decoy_one', 'ctor')        # every agent stacks decoys on one host: 257 processes on it at the end
        record(322, 500, 'decoy', 'reset')
        record(323, 500, 'decoy_restore', 'ctor')
        record(324, 500, 'block_allow', 'ctor')
        record(325, 500, 'remove', 'reset')
        record(326, 500, 'restore', 'ctor')
        record(327, 500, 'mix', 'ctor')
        record(328, 300, 'decoy', 'ctor', red='random')
        record(329, 300, 'mix', 'reset', red='discovery')
        sys.exit(0)
    # BASELINE config 1 (SleepAgent blue, FSM red, 500 steps) + seeded random blue; regression seeds of
    # CybORG/Tests/test_cc4/test_heuristic_agents.py
    record(123, 500, 'sleep', 'ctor')
    record(7, 500, 'sleep', 'ctor')
    record(123, 500, 'random', 'ctor')
    record(100, 500, 'random', 'reset')
    record(3, 500, 'random', 'ctor', msgs=True)
    record(6065, 500, 'random', 'reset')
    record(5712, 100, 'random', 'ctor')
    record(87, 30, 'random', 'reset')
    record(9283, 6, 'sleep', 'ctor')
