"""gymnasium.spaces when available, otherwise the three tiny space classes the CC4 wrappers need
(gymnasium is not installed in the build image)."""
import numpy as np

try:  # pragma: no cover
    from gymnasium.spaces import Discrete, MultiDiscrete, MultiBinary  # type: ignore
except Exception:  # noqa: BLE001
    class Discrete:
        def __init__(self, n, seed=None):
            self.n = int(n)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return int(self._rng.integers(self.n))

        def contains(self, x):
            return isinstance(x, (int, np.integer)) and 0 <= int(x) < self.n

        def __eq__(self, other):
            return isinstance(other, Discrete) and other.n == self.n

        def __repr__(self):
            return f"Discrete({self.n})"

    class MultiDiscrete:
        def __init__(self, nvec, seed=None):
            self.nvec = np.asarray(nvec, dtype=np.int64)
            self.shape = self.nvec.shape
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return (self._rng.random(self.nvec.shape) * self.nvec).astype(np.int64)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.nvec.shape and bool(np.all((x >= 0) & (x < self.nvec)))

        def __len__(self):
            return len(self.nvec)

        def __eq__(self, other):
            return isinstance(other, MultiDiscrete) and np.array_equal(other.nvec, self.nvec)

        def __repr__(self):
            return f"MultiDiscrete({self.nvec.tolist()})"

    class MultiBinary:
        def __init__(self, n, seed=None):
            self.n = int(n)
            self.shape = (self.n,)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.integers(0, 2, size=self.n).astype(bool)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == (self.n,) and bool(np.all((x == 0) | (x == 1)))

        def __repr__(self):
            return f"MultiBinary({self.n})"
