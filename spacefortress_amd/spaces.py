"""Minimal stand-ins for gym.spaces.{Box, Discrete} -- only what the reference's callers read
(`.shape`, `.n`, `.dtype`, `.sample()`; rl/train.py:18,37-41).  When gym is importable the real
classes are used instead, so isinstance checks in user code keep working."""
import numpy as np

try:  # pragma: no cover - gym is not installed in the build image
    from gym.spaces import Box, Discrete  # type: ignore
except Exception:

    class Discrete:
        def __init__(self, n):
            self.n = int(n)
            self.shape = ()
            self.dtype = np.dtype(np.int64)

        def sample(self):
            return int(np.random.randint(self.n))

        def contains(self, x):
            return 0 <= int(x) < self.n

        def __repr__(self):
            return "Discrete(%d)" % self.n

    class Box:
        def __init__(self, low, high, shape=None, dtype=np.float32):
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.dtype = np.dtype(dtype)
            self.low = np.full(self.shape, low, dtype=np.float64)
            self.high = np.full(self.shape, high, dtype=np.float64)

        def sample(self):
            return np.random.uniform(-1, 1, self.shape).astype(self.dtype)

        def __repr__(self):
            return "Box%s" % (self.shape,)
