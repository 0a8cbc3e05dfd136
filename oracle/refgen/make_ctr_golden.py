"""Golden trajectories of the COUNTER (Philox) mode's step, recorded from the REAL reference (CybORG v4 at /root/reference)
running under oracle/refgen/philox_proxy.PhiloxProxy: scenario from the numpy stream of `seed` (CybORG(seed=proxy) +
wrapper.reset()), dynamics on the counter streams of `key` (episode word 1: what cc4_set_seed leaves).  Data only: seed, key,
per-step blue action indices, and the reference's outputs (flat observations, team reward, done, action mask).

usage: python make_ctr_golden.py [long]   # writes tests/golden/ctrstep_*.npz (long: the two 1000-step episodes)
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from compare import RED, GREEN
from compare_ctr import make_pair
from blue_policies import BluePolicy, KINDS

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden')


def flat(obs):
    return np.concatenate([obs[f'blue_agent_{b}'] for b in range(5)]).astype(np.uint8)


def record(seed, steps, blue, red='fsm', green='enterprise', key=None, msgs=False):
    key = seed if key is None else key
    env, w, proxy, H, pol, obs, info = make_pair(seed, steps, blue, red, green, key)
    arng = np.random.default_rng(seed ^ 0xB10E)
    bpol = BluePolicy(blue, {f'blue_agent_{b}': w.action_labels(f'blue_agent_{b}') for b in range(5)}, seed) if blue in KINDS else None
    A = np.full((steps, 5), -1, np.int16)
    O = np.zeros((steps + 1, 578), np.uint8)
    R = np.zeros(steps, np.float32)
    D = np.zeros(steps, np.uint8)
    M = np.zeros((steps, 5, 8), np.uint8)
    mask = np.concatenate([np.array(info[f'blue_agent_{b}']['action_mask'], np.uint8) for b in range(5)])
    O[0] = flat(obs)
    for t in range(steps):
        acts = {}
        if blue == 'random':
            A[t] = [arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(242)]
            acts = {f'blue_agent_{b}': int(A[t, b]) for b in range(5)}
        elif bpol is not None:
            A[t] = bpol.act(t)
            acts = {f'blue_agent_{b}': int(A[t, b]) for b in range(5)}
        messages = None
        if msgs:
            M[t] = arng.integers(0, 2, size=(5, 8))
            messages = {f'blue_agent_{b}': M[t, b].astype(bool) for b in range(5)}
        proxy.begin_step(env.environment_controller.step_count)
        obs, rew, term, trunc, info = w.step(acts, messages=messages)
        O[t + 1] = flat(obs)
        R[t] = rew['blue_agent_0']
        assert len(set(rew.values())) == 1
        D[t] = term['blue_agent_0']
    pol_s = ('' if red == 'fsm' else f'_red{red}') + ('' if green == 'enterprise' else f'_green{green}')
    name = f"ctrstep_seed{seed}_{blue}_{steps}{'_msg' if msgs else ''}{pol_s}.npz"
    np.savez_compressed(os.path.join(OUT, name), seed=np.int64(seed), key=np.uint64(key), steps=np.int32(steps), actions=A,
                        obs_bits=np.packbits((O > 0).astype(np.uint8), axis=1), phase_cols=np.array([0, 92, 184, 276, 368], np.int32),
                        obs_phase_vals=O[:, [0, 92, 184, 276, 368]].copy(), reward=R, done=D, mask=mask,
                        messages=M if msgs else np.zeros(0, np.uint8), red_policy=np.int32(RED[red][1]), green_policy=np.int32(GREEN[green][1]),
                        blue_policy=np.int32(blue == 'builtin'), numpy_version=np.bytes_(np.__version__),
                        proxy_calls=np.bytes_(repr(sorted(proxy.calls.items()))))
    print(name, 'sum reward', float(R.sum()), 'proxy calls', dict(sorted(proxy.calls.items())), flush=True)


if __name__ == '__main__':
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'long':      # EnterpriseScenarioGenerator(steps=1000): the containers sized from the episode length
        record(777, 1000, 'decoy_one')
        record(778, 1000, 'random', red='random')
        sys.exit(0)
    record(123, 500, 'random')
    record(7, 500, 'sleep', key=99991)
    record(3, 500, 'random', msgs=True)
    record(327, 500, 'mix')
    record(321, 500, 'decoy_one')
    record(326, 500, 'restore', key=(1 << 64) - 59)          # a key with a high word
    record(44, 500, 'random', red='random')
    record(41, 500, 'random', red='discovery')
    record(87, 500, 'builtin', red='random')
    record(43, 200, 'random', red='discovery', green='sleep')
