"""Observation/action spaces.  gymnasium's ``Box`` is used when gymnasium is installed (so that SB3 / RLlib /
``check_env`` see the real thing); otherwise a minimal stand-in with the same attributes."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from gymnasium.spaces import Box  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # gymnasium is not installed in the build image
    HAVE_GYMNASIUM = False

    class Box:  # type: ignore
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            self.shape = tuple(shape) if shape is not None else np.shape(low)
            self.low = np.full(self.shape, low, dtype=self.dtype)
            self.high = np.full(self.shape, high, dtype=self.dtype)
            self._rng = np.random.default_rng(seed)

        def sample(self):
            return self._rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

        def seed(self, seed=None):
            self._rng = np.random.default_rng(seed)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self.shape}, {self.dtype})"
