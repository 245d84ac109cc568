"""Minimal `gymnasium.spaces.Box` stand-in (gymnasium is not a dependency of this package).

If gymnasium is importable its Box is used, so that wrappers/learners written against it see
the real type; otherwise this shim provides the attributes the reference's callers touch
(`shape`, `low`, `high`, `dtype`, `sample()`, `contains()`; reference agents/agent.py:153-161,
examples/benchmark.py:100).
"""
import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box  # type: ignore
except Exception:  # gymnasium absent

    class Box:
        def __init__(self, low, high, dtype=np.float32, seed=None):
            self.low = np.asarray(low, dtype=dtype)
            self.high = np.asarray(high, dtype=dtype)
            self.shape = self.low.shape
            self.dtype = np.dtype(dtype)
            self._rng = np.random.default_rng(seed)

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def sample(self):
            lo = np.where(np.isfinite(self.low), self.low, -1.0)
            hi = np.where(np.isfinite(self.high), self.high, 1.0)
            return self._rng.uniform(lo, hi).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
