"""Import shim that lets the *reference* CybORG (Python, /root/reference) run in the build
container, which lacks gym / gymnasium / pygame / prettytable / ray.

TEST INFRASTRUCTURE ONLY.  Used by oracle/refgen/*.py to (a) validate the C++ restatement in
oracle/ against the real reference and (b) emit the golden fixtures under tests/golden/.
/root/reference never travels to the GPU box, so nothing under tests/ -m gpu, smoke() or
bench.py imports this module.

`np_random(seed)` restates gym 0.26.2 `gym.utils.seeding.np_random`:
    Generator(PCG64(SeedSequence(seed)))
"""
import sys
import types
import numpy as np

REFERENCE_ROOT = '/root/reference'


def _mod(name):
    m = types.ModuleType(name)
    sys.modules[name] = m
    return m


def install():
    if 'gym' in sys.modules and getattr(sys.modules['gym'], '_cc4_shim', False):
        return
    sys.dont_write_bytecode = True

    class Space:
        def __init__(self, *a, **k):
            self._np_random = None

    class Discrete(Space):
        def __init__(self, n, *a, **k):
            super().__init__()
            self.n = int(n)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

    class MultiDiscrete(Space):
        def __init__(self, nvec, *a, **k):
            super().__init__()
            self.nvec = np.asarray(nvec)

        def __len__(self):
            return len(self.nvec)

    class MultiBinary(Space):
        def __init__(self, n, *a, **k):
            super().__init__()
            self.n = n

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == (self.n,) and bool(np.all((x == 0) | (x == 1)))

        def sample(self):
            rng = self._np_random or np.random.default_rng()
            return rng.integers(0, 2, size=self.n).astype(bool)

    class Box(Space):
        def __init__(self, *a, **k):
            super().__init__()

    class Dict_(Space, dict):
        def __init__(self, d=None, *a, **k):
            Space.__init__(self)
            dict.__init__(self, d or {})

    def np_random(seed=None):
        ss = np.random.SeedSequence(seed)
        return np.random.Generator(np.random.PCG64(ss)), ss.entropy

    for root in ('gym', 'gymnasium'):
        g = _mod(root)
        g._cc4_shim = True
        sp = _mod(root + '.spaces')
        for c in (Space, Discrete, MultiDiscrete, MultiBinary, Box):
            setattr(sp, c.__name__, c)
        sp.Dict = Dict_
        g.spaces = sp
        g.Space = Space
        g.Env = type('Env', (), {})
        u = _mod(root + '.utils')
        sd = _mod(root + '.utils.seeding')
        sd.np_random = np_random
        sd.RandomNumberGenerator = np.random.Generator
        u.seeding = sd
        g.utils = u
        v = _mod(root + '.vector')
        vu = _mod(root + '.vector.utils')
        vs = _mod(root + '.vector.utils.spaces')
        v.utils = vu
        vu.spaces = vs
        g.vector = v
    _mod('pygame')
    pt = _mod('prettytable')
    pt.PrettyTable = type('PrettyTable', (), {'__init__': lambda self, *a, **k: None})
    ray = _mod('ray')
    rl = _mod('ray.rllib')
    re_ = _mod('ray.rllib.env')
    mae = _mod('ray.rllib.env.multi_agent_env')
    mae.MultiAgentEnv = type('MultiAgentEnv', (), {})
    ray.rllib = rl
    rl.env = re_
    re_.multi_agent_env = mae
    pm = _mod('pytest_mock')
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)


install()
