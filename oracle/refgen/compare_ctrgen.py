"""Differential test of the COUNTER (Philox) mode INCLUDING ITS SCENARIO GENERATION: the reference CybORG (this container only),
built and reset under oracle/refgen/philox_proxy.PhiloxProxy armed for generation, against the CPU oracle's counter-mode reset
(env_reset_counter_mode) and step.  Both sides: scenario #1 at construction (key, episode word 1), scenario #2 at the wrapper's
reset() (episode word 2), `steps` steps of random blue actions, then -- the episode over -- a third scenario (reset(seed=None):
what the kernels' in-kernel autoreset does) and a few more steps.  After every reset and step: flat observations, action mask,
reward, done, full canonical state dump.

usage: python compare_ctrgen.py <key> [steps] [red: fsm|sleep|discovery|random] [green: enterprise|sleep] [more_steps]"""
import sys, os, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(__file__))
import ref_shim  # noqa
from compare import lib, canon_ref, RED, GREEN
from ref_dump import dump
from philox_proxy import PhiloxProxy
from CybORG import CybORG
from CybORG.Simulator.Scenarios import EnterpriseScenarioGenerator
from CybORG.Simulator.Actions import Action
from CybORG.Agents import SleepAgent
from CybORG.Agents.Wrappers import BlueFlatWrapper


def flat(obs):
    return np.concatenate([obs[f'blue_agent_{b}'] for b in range(5)]).astype(np.int32)


class Pair:
    def __init__(self, key, steps, red='fsm', green='enterprise'):
        sg = EnterpriseScenarioGenerator(blue_agent_class=SleepAgent, green_agent_class=GREEN[green][0], red_agent_class=RED[red][0], steps=steps)
        self.pol = RED[red][1] | (0x10 if GREEN[green][1] else 0)
        self.steps = steps
        self.proxy = PhiloxProxy(0, key)
        self.proxy.arm(Action, generation=True)
        self.proxy.begin_reset()
        self.env = CybORG(sg, seed=self.proxy)                      # scenario #1 (episode word 1)
        self.w = BlueFlatWrapper(self.env)
        self.H = ctypes.c_void_p(lib.cc4o_create2(1, steps))
        lib.cc4o_reset(self.H, 0, ctypes.c_uint64(key), 1, steps, 0, self.pol)
        self.buf = ctypes.create_string_buffer(1 << 20)

    def reset(self):
        self.proxy.begin_reset()
        obs, info = self.w.reset()
        lib.cc4o_reset(self.H, 0, 0, 1, self.steps, 1, self.pol)
        return obs, info

    def step(self, a):
        self.proxy.begin_step(self.env.environment_controller.step_count)
        obs, rew, term, trunc, info = self.w.step({f'blue_agent_{b}': int(a[b]) for b in range(5)})
        lib.cc4o_step(self.H, 0, np.ascontiguousarray(a, np.int32).ctypes.data_as(ctypes.c_void_p), None)
        return obs, rew, term

    def oracle_obs(self):
        o = np.zeros(578, np.int32)
        lib.cc4o_obs(self.H, 0, o.ctypes.data_as(ctypes.c_void_p))
        return o

    def oracle_mask(self):
        m = np.zeros(570, np.uint8)
        lib.cc4o_mask(self.H, 0, m.ctypes.data_as(ctypes.c_void_p))
        return m

    def dumps(self):
        n = lib.cc4o_dump(self.H, 0, self.buf, len(self.buf))
        return self.buf.raw[:n].decode(), canon_ref(dump(self.env))

    def check(self, tag, obs, info=None, rew=None, term=None):
        ok = True
        if not np.array_equal(self.oracle_obs(), flat(obs)):
            print(tag, 'OBS MISMATCH at', np.nonzero(self.oracle_obs() != flat(obs))[0][:20]); ok = False
        if info is not None:
            rm = np.concatenate([np.array(info[f'blue_agent_{b}']['action_mask'], np.uint8) for b in range(5)])
            if not np.array_equal(self.oracle_mask(), rm):
                print(tag, 'MASK MISMATCH'); ok = False
        if rew is not None and abs(lib.cc4o_reward(self.H, 0) - rew['blue_agent_0']) > 1e-6:
            print(tag, 'REWARD MISMATCH', lib.cc4o_reward(self.H, 0), rew['blue_agent_0']); ok = False
        if term is not None and bool(lib.cc4o_done(self.H, 0)) != bool(term['blue_agent_0']):
            print(tag, 'DONE MISMATCH'); ok = False
        mine, ref = self.dumps()
        if mine != ref:
            ok = False
            k = 0
            for a, b in zip(mine.split('\n'), ref.split('\n')):
                if a != b and k < 12:
                    print(tag, 'STATE DIFF\n  mine:', a, '\n  ref :', b); k += 1
        if lib.cc4o_err(self.H, 0):
            print(tag, 'ERR FLAGS', hex(lib.cc4o_err(self.H, 0))); ok = False
        return ok


def run(key, steps=100, red='fsm', green='enterprise', more=25, verbose=True):
    p = Pair(key, steps, red, green)
    mine, ref = p.dumps()
    if mine != ref:
        k = 0
        for a, b in zip(mine.split('\n'), ref.split('\n')):
            if a != b and k < 12:
                print('scenario #1 STATE DIFF\n  mine:', a, '\n  ref :', b); k += 1
        return -2
    obs, info = p.reset()
    if not p.check('scenario #2', obs, info):
        return -1
    arng = np.random.default_rng(key ^ 0xB10E)
    for t in range(steps):
        a = np.array([arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(242)], np.int32)
        obs, rew, term = p.step(a)
        if not p.check(f'step {t}', obs, None, rew, term):
            return t
    obs, info = p.reset()                                            # the episode is over: scenario #3 on the running key
    if not p.check('scenario #3', obs, info):
        return -3
    for t in range(more):
        a = np.array([arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(82), arng.integers(242)], np.int32)
        obs, rew, term = p.step(a)
        if not p.check(f'step {steps}+{t}', obs, None, rew, term):
            return steps + t
    if verbose:
        print('OK key', key, red, green, 'calls', dict(sorted(p.proxy.calls.items())))
    return None


if __name__ == '__main__':
    key = int(sys.argv[1])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    red = sys.argv[3] if len(sys.argv) > 3 else 'fsm'
    green = sys.argv[4] if len(sys.argv) > 4 else 'enterprise'
    more = int(sys.argv[5]) if len(sys.argv) > 5 else 25
    sys.exit(0 if run(key, steps, red, green, more) is None else 1)
