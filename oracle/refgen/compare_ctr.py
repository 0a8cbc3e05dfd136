"""Differential test of the COUNTER (Philox) mode's step: the reference CybORG (this container only) driven by
oracle/refgen/philox_proxy.PhiloxProxy against the CPU oracle in its counter mode, step by step, full state dump.

Both sides generate the scenario from the numpy stream of `seed` (CybORG(seed=proxy): proxy delegates to Generator(PCG64);
oracle: env_reset in mode 0, twice, as CybORG.__init__ + wrapper.reset() do), then the dynamics run on the counter streams
of key `key` (reference: proxy armed; oracle: cc4o_set_seed(key, mode 1) == cc4_set_seed of the C ABI).

usage: python compare_ctr.py <seed> [steps] [blue: sleep|random|builtin|<blue_policies.KINDS>] [red: fsm|sleep|discovery|random] [green: enterprise|sleep] [key]"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from compare import lib, canon_ref, RED, GREEN
from ref_dump import dump
from blue_policies import BluePolicy, KINDS
from philox_proxy import PhiloxProxy
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Simulator.Actions import Action
from CybORG.Agents import SleepAgent, cc4BlueRandomAgent
from CybORG.Agents.Wrappers import BlueFlatWrapper

lib.cc4o_set_seed.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint64, ctypes.c_int]


def make_pair(seed, steps, blue, red, green, key):
    """(reference wrapper, proxy, oracle handle, policy bits) with the scenario generated and the counter streams armed."""
    sg = EnterpriseScenarioGenerator(blue_agent_class=cc4BlueRandomAgent if blue == 'builtin' else SleepAgent,
                                     green_agent_class=GREEN[green][0], red_agent_class=RED[red][0], steps=steps)
    pol = RED[red][1] | (0x10 if GREEN[green][1] else 0) | (0x20 if blue == 'builtin' else 0)
    proxy = PhiloxProxy(seed, key)
    env = CybORG(sg, seed=proxy)
    w = BlueFlatWrapper(env)
    obs, info = w.reset()                                  # second scenario from the running numpy stream
    H = ctypes.c_void_p(lib.cc4o_create2(1, steps))
    lib.cc4o_reset(H, 0, ctypes.c_uint64(seed), 0, steps, 0, pol)
    lib.cc4o_reset(H, 0, ctypes.c_uint64(seed), 0, steps, 1, pol)
    lib.cc4o_set_seed(H, 0, ctypes.c_uint64(key), 1)       # dynamics: counter streams of `key` (episode word 1)
    proxy.arm(Action)
    return env, w, proxy, H, pol, obs, info


def run(seed, steps=500, blue='random', red='fsm', green='enterprise', key=None, verbose=True, max_steps=None):
    key = seed if key is None else key
    env, w, proxy, H, pol, obs, info = make_pair(seed, steps, blue, red, green, key)
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

    if not check('reset', obs):
        return -1
    total = 0.0
    for t in range(max_steps or steps):
        if blue in ('sleep', 'builtin'):
            acts = {}
            a = np.full(5, -1, np.int32)
        elif bpol is not None:
            a = bpol.act(t)
            acts = {f'blue_agent_{b}': int(a[b]) for b in range(5)}
        else:
            a = np.array([arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(242)], np.int32)
            acts = {f'blue_agent_{b}': int(a[b]) for b in range(5)}
        proxy.begin_step(env.environment_controller.step_count)
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
        print('OK seed', seed, 'key', key, blue, red, green, 'total reward', total, 'calls', dict(sorted(proxy.calls.items())))
    return None


if __name__ == '__main__':
    seed = int(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 500
    blue = sys.argv[3] if len(sys.argv) > 3 else 'random'
    red = sys.argv[4] if len(sys.argv) > 4 else 'fsm'
    green = sys.argv[5] if len(sys.argv) > 5 else 'enterprise'
    key = int(sys.argv[6]) if len(sys.argv) > 6 else None
    r = run(seed, steps, blue, red, green, key)
    sys.exit(0 if r is None else 1)
