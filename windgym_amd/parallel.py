"""Multi-GPU: the env axis shards embarrassingly (one process per GPU, one handle per process); the only
exchange is the episode-metric vector — the distributed form of ``RecordEpisodeVals``
(wrappers/recordEpisodeVals.py:31-64) and of the ``WindFarmMonitor`` callback
(examples/longer_steps_example.py:39-124): ONE all-reduce(sum) of 8 floats over RCCL ("nccl" backend on
ROCm; "gloo" in the CPU tests)."""
from __future__ import annotations

METRIC_NAMES = ("ep_return_sum", "ep_length_sum", "ep_mean_power_sum", "n_episodes", "step_reward_sum",
                "farm_power_sum", "base_power_sum", "n_steps")


def shard_range(n_envs_total: int, rank: int, world: int):
    """Contiguous shard of the global env axis owned by `rank` (rank g owns [g*B/G, (g+1)*B/G))."""
    per = n_envs_total // world
    rem = n_envs_total % world
    lo = rank * per + min(rank, rem)
    return lo, lo + per + (1 if rank < rem else 0)


def global_seeds(base_seed: int, n_envs_total: int, rank: int, world: int):
    """Seeds keyed by the *global* env index, so results do not depend on the number of GPUs."""
    import numpy as np
    lo, hi = shard_range(n_envs_total, rank, world)
    return base_seed + np.arange(lo, hi, dtype=np.uint64)


def reduce_metric_vector(vec):
    """all-reduce(sum) of the 8-float partial-sum vector (in place) when a process group exists."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(vec, op=dist.ReduceOp.SUM)
    return vec


def derive(vec):
    """Turn the reduced sums into the quantities the reference logs."""
    v = [float(x) for x in vec]
    out = dict(zip(METRIC_NAMES, v))
    n_ep = max(out["n_episodes"], 1.0)
    n_st = max(out["n_steps"], 1.0)
    out["mean_episode_return"] = out["ep_return_sum"] / n_ep
    out["mean_episode_length"] = out["ep_length_sum"] / n_ep
    out["mean_episode_power"] = out["ep_mean_power_sum"] / n_ep      # RecordEpisodeVals.mean_power_queue
    out["mean_step_reward"] = out["step_reward_sum"] / n_st
    out["mean_farm_power"] = out["farm_power_sum"] / n_st
    out["mean_base_power"] = out["base_power_sum"] / n_st
    return out


class ShardedMetrics:
    """Episode metrics of the whole (sharded) batch."""

    def __init__(self, batch):
        self.batch = batch

    def all_reduce(self, reset_after=True):
        vec = self.batch.metrics(reset_after=reset_after).clone()
        return derive(reduce_metric_vector(vec).cpu())
