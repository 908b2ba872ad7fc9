"""windgym_amd — MI355X-native batched implementation of WindGym's step() transition.

Public names mirror the reference package (WindGym/__init__.py): ``WindFarmEnv``, ``FarmEval``,
``WindFarmEnvMulti``; plus the native batched ``WindFarmVecEnv`` and the ``V80`` tabular turbine.
"""
from .turbine import V80, TabularTurbine  # noqa: F401
from .config import EnvConfig  # noqa: F401


def __getattr__(name):   # lazy: importing the env classes pulls in torch
    if name in ("WindFarmEnv", "FarmEval", "WindFarmEnvMulti", "WindFarmVecEnv", "RecordEpisodeVals"):
        from . import envs
        return getattr(envs, name)
    if name == "HipBatch":
        from .binding import HipBatch
        return HipBatch
    if name in ("AgentEval", "eval_sweep"):
        from . import evaluate
        return getattr(evaluate, name)
    if name in ("ConstantAgent", "RandomAgent", "GreedyAgent", "BaseAgent"):
        from . import agents
        return getattr(agents, name)
    if name in ("PyWakeAgent", "SteadyStateYawAgent"):
        from . import steady
        return getattr(steady, name)
    raise AttributeError(name)
