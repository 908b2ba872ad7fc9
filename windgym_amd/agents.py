"""Scripted agents with the reference's `predict()` protocol, batched (WindGym/Agents/*.py).

`predict(obs, deterministic=...) -> (action, None)`; actions are float32 arrays `[n_envs, n_turb]` in [-1, 1]
(or `[n_turb]` for the single-env classes).  `scale_yaw` is BaseAgent.scale_yaw (Agents/BaseAgent.py:17-23):
a yaw target in degrees -> the action that the "wind" ActionMethod maps back to that target
(Wind_Farm_Env.py:841-843)."""
from __future__ import annotations

import numpy as np


class BaseAgent:
    def __init__(self, yaw_max=45, yaw_min=-45):
        self.yaw_max, self.yaw_min = yaw_max, yaw_min

    def scale_yaw(self, yaws):
        return (np.asarray(yaws, dtype=np.float64) - self.yaw_min) / (self.yaw_max - self.yaw_min) * 2 - 1

    def predict(self, *args, **kwargs):
        raise NotImplementedError


class ConstantAgent(BaseAgent):
    """Agents/ConstantAgent.py: hold predefined yaw angles (degrees)."""

    def __init__(self, yaw_angles, yaw_max=45, yaw_min=-45):
        super().__init__(yaw_max, yaw_min)
        self.UseEnv = True
        self.yaw_angles = np.asarray(yaw_angles, dtype=np.float64)

    def predict(self, *args, **kwargs):
        return self.scale_yaw(self.yaw_angles).astype(np.float32), None


class RandomAgent(BaseAgent):
    """Agents/RandomAgent.py: uniform random actions."""

    def __init__(self, shape, seed=None):
        super().__init__()
        self.shape = tuple(np.atleast_1d(shape))
        self._rng = np.random.default_rng(seed)

    def predict(self, *args, **kwargs):
        return self._rng.uniform(-1, 1, size=self.shape).astype(np.float32), None


class GreedyAgent(BaseAgent):
    """Agents/GreedyAgent.py: steer every turbine back into its local (or the global) wind direction with the
    baseline controllers' logic (BasicControllers.py:10-73), expressed as a yaw target for the "wind" method."""

    def __init__(self, type="local", yaw_max=45, yaw_min=-45, yaw_step=1, env=None):
        super().__init__(yaw_max, yaw_min)
        self.UseEnv = True
        self.env, self.type, self.yaw_step = env, type, yaw_step

    def predict(self, *args, **kwargs):
        env = self.env
        batch = getattr(env, "batch", None) or getattr(env, "_batch")
        yaw = batch.info("yaw_agent").cpu().numpy().astype(np.float64)
        if self.type == "local":
            uvw = batch.info("rotor_uvw_agent").cpu().numpy().astype(np.float64)
            off = np.rad2deg(np.arctan(uvw[..., 1] / uvw[..., 0])) - yaw
            goal = yaw + np.sign(off) * np.minimum(np.abs(off), self.yaw_step)
        else:
            goal = yaw - np.sign(yaw) * np.minimum(np.abs(yaw), self.yaw_step)
        a = self.scale_yaw(goal).astype(np.float32)
        return (a if hasattr(env, "batch") else a[0]), None
