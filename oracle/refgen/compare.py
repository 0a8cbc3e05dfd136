"""Differential test: reference CybORG (this container only) vs the CPU oracle, step by step.
usage: python compare.py <seed> [steps] [blue: sleep|random|builtin (= blue_agent_class=cc4BlueRandomAgent, no actions submitted)|<blue_policies.KINDS>] [init: ctor|reset]"""
import sys, os, ctypes, re
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from ref_dump import dump
from blue_policies import BluePolicy, KINDS
from blue_policies import BluePolicy, KINDS
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Agents import SleepAgent, EnterpriseGreenAgent, FiniteStateRedAgent, DiscoveryFSRed, RandomSelectRedAgent, cc4BlueRandomAgent
RED = {'fsm': (FiniteStateRedAgent, 0), 'sleep': (SleepAgent, 1), 'discovery': (DiscoveryFSRed, 2), 'random': (RandomSelectRedAgent, 3)}
GREEN = {'enterprise': (EnterpriseGreenAgent, 0), 'sleep': (SleepAgent, 1)}
from CybORG.Agents.Wrappers import BlueFlatWrapper

RED_QT = {'DiscoverRemoteSystems': 0, 'AggressiveServiceDiscovery': 1, 'StealthServiceDiscovery': 2, 'DiscoverDeception': 3,
          'ExploitRemoteService': 4, 'PrivilegeEscalate': 5, 'Impact': 6, 'DegradeServices': 7, 'Withdraw': 8, 'Sleep': 9,
          'InvalidAction': 10}

lib = ctypes.CDLL(os.environ.get('CC4_ORACLE_LIB') or os.path.join(os.path.dirname(__file__), '..', 'liboracle.so'))
lib.cc4o_create.restype = ctypes.c_void_p
lib.cc4o_create2.restype = ctypes.c_void_p
lib.cc4o_reward.restype = ctypes.c_float
for f in ('cc4o_reset', 'cc4o_step', 'cc4o_obs', 'cc4o_reward', 'cc4o_done', 'cc4o_err', 'cc4o_mask', 'cc4o_rng_state', 'cc4o_dump'):
    getattr(lib, f).argtypes = None


def canon_ref(txt):
    return re.sub(r'qt (\w+)', lambda m: 'qt ' + str(RED_QT.get(m.group(1), m.group(1))), txt)


def run(seed, steps=500, blue='sleep', init='ctor', verbose=True, max_steps=None, red='fsm', green='enterprise'):
    sg = EnterpriseScenarioGenerator(blue_agent_class=cc4BlueRandomAgent if blue == 'builtin' else SleepAgent,
                                     green_agent_class=GREEN[green][0], red_agent_class=RED[red][0], steps=steps)
    pol = RED[red][1] | (0x10 if GREEN[green][1] else 0) | (0x20 if blue == 'builtin' else 0)
    env = CybORG(sg, seed=seed)
    w = BlueFlatWrapper(env)
    H = ctypes.c_void_p(lib.cc4o_create2(1, steps))
    lib.cc4o_reset(H, 0, ctypes.c_uint64(seed), 0, steps, 0, pol)
    if init == 'ctor':
        obs, info = w.reset()
        lib.cc4o_reset(H, 0, ctypes.c_uint64(seed), 0, steps, 1, pol)
    else:
        obs, info = w.reset(seed=seed + 1)
        lib.cc4o_reset(H, 0, ctypes.c_uint64(seed + 1), 0, steps, 0, pol)
    arng = np.random.default_rng(seed ^ 0xB10E)
    bpol = BluePolicy(blue, {f'blue_agent_{b}': w.action_labels(f'blue_agent_{b}') for b in range(5)}, seed) if blue in KINDS else None
    buf = ctypes.create_string_buffer(1 << 20)

    def check(tag, obs, rew=None, done=None):
        ok = True
        o = np.zeros(578, np.int32)
        lib.cc4o_obs(H, 0, o.ctypes.data_as(ctypes.c_void_p))
        ro = np.concatenate([obs[f'blue_agent_{b}'] for b in range(5)]).astype(np.int32)
        if not np.array_equal(o, ro):
            print(tag, 'OBS MISMATCH at', np.nonzero(o != ro)[0][:20]); ok = False
        st = env.environment_controller.np_random.bit_generator.state
        v = st['state']['state']
        rs = (ctypes.c_uint64 * 7)()
        lib.cc4o_rng_state(H, 0, rs)
        if (rs[0] << 64 | rs[1]) != v or rs[4] != st['has_uint32'] or (rs[4] and rs[5] != st['uinteger']):
            print(tag, 'RNG MISMATCH'); ok = False
        if rew is not None:
            r = lib.cc4o_reward(H, 0)
            if abs(r - rew['blue_agent_0']) > 1e-6:
                print(tag, 'REWARD MISMATCH', r, rew['blue_agent_0']); ok = False
            if bool(lib.cc4o_done(H, 0)) != bool(done['blue_agent_0']):
                print(tag, 'DONE MISMATCH'); ok = False
        n = lib.cc4o_dump(H, 0, buf, len(buf))
        mine = buf.raw[:n].decode()
        ref = canon_ref(dump(env))
        if mine != ref:
            ok = False
            for a, b in zip(mine.split('\n'), ref.split('\n')):
                if a != b:
                    print(tag, 'STATE DIFF\n  mine:', a, '\n  ref :', b)
        err = lib.cc4o_err(H, 0)
        if err:
            print(tag, 'ERR FLAGS', hex(err)); ok = False
        return ok

    m = np.zeros(570, np.uint8)
    lib.cc4o_mask(H, 0, m.ctypes.data_as(ctypes.c_void_p))
    rm = np.concatenate([np.array(info[f'blue_agent_{b}']['action_mask'], np.uint8) for b in range(5)])
    if not np.array_equal(m, rm):
        print('MASK MISMATCH', np.nonzero(m != rm)[0]); return -1
    if not check('reset', obs):
        return -1
    total = 0.0
    nst = max_steps or steps
    for t in range(nst):
        if blue in ('sleep', 'builtin'):
            acts = {}
            a = np.full(5, -1, np.int32)
        elif bpol is not None:
            a = bpol.act(t)
            acts = {f'blue_agent_{b}': int(a[b]) for b in range(5)}
        else:
            a = np.array([arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(242)], np.int32)
            acts = {f'blue_agent_{b}': int(a[b]) for b in range(5)}
        obs, rew, term, trunc, info = w.step(acts)
        lib.cc4o_step(H, 0, a.ctypes.data_as(ctypes.c_void_p), None)
        total += rew['blue_agent_0']
        if not check(f'step {t}', obs, rew, term):
            if verbose:
                print('actions', a.tolist())
                for r in range(6):
                    print(' red', r, env.environment_controller.action.get(f'red_agent_{r}'))
            return t
    if verbose:
        print('OK seed', seed, 'total reward', total)
    return None


if __name__ == '__main__':
    seed = int(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    blue = sys.argv[3] if len(sys.argv) > 3 else 'sleep'
    init = sys.argv[4] if len(sys.argv) > 4 else 'ctor'
    red = sys.argv[5] if len(sys.argv) > 5 else 'fsm'
    green = sys.argv[6] if len(sys.argv) > 6 else 'enterprise'
    run(seed, steps, blue, init, red=red, green=green)
