"""SURVEY.md §8a row a15 — RecordEpisodeVals (wrappers/recordEpisodeVals.py:31-64), the host form of the episode
metrics that get all-reduced across GPUs.

Chain of evidence:
  reference wrapper (golden vector, tests/golden/record_episode_vals.npz, recorded from the reference's own file)
    == windgym_amd.envs.RecordEpisodeVals on the same stream                       (this file, exact)
    == the oracle's device-style accumulators (ep_mean_power_sum, ...)             (this file, 1e-12)
    == the HIP kernels' accumulators                                               (tests/test_gpu_envs.py, fp32).

The reference runs under gymnasium's NEXT-step autoreset (the step after a done is a reset step that the wrapper does
not count); the batched env resets in the SAME step.  The two streams differ exactly by those reset rows: dropping them
from the golden stream gives the same-step stream, and the per-episode means must be identical.
"""
import os

import numpy as np

from helpers import GOLDEN

G = np.load(os.path.join(GOLDEN, "record_episode_vals.npz"))


class _ScriptedVecEnv:
    """Same-step-autoreset vector env double fed from the golden tables with the reset rows removed."""
    as_torch = False

    def __init__(self):
        self.num_envs = G["power"].shape[1]
        # per env: the rows that are real steps
        keep = ~G["is_reset_row"]
        self.rows = [np.nonzero(keep[:, b])[0] for b in range(self.num_envs)]
        self.n_steps = min(len(r) for r in self.rows)
        self.t = -1

    def reset(self, **kw):
        self.t = -1
        return np.zeros((self.num_envs, 1)), {}

    def step(self, actions):
        self.t += 1
        idx = [r[self.t] for r in self.rows]
        b = np.arange(self.num_envs)
        return (np.zeros((self.num_envs, 1)), G["reward"][idx, b], np.zeros(self.num_envs, dtype=bool), G["done"][idx, b],
                {"Power agent": G["power"][idx, b]})


def test_host_wrapper_reproduces_the_reference_wrapper():
    from windgym_amd.envs import RecordEpisodeVals
    env = _ScriptedVecEnv()
    w = RecordEpisodeVals(env, buffer_length=10000)
    w.reset()
    for _ in range(env.n_steps):
        w.step(None)
    # per env, the episodes complete in the same order in both streams; across envs the interleaving differs (the
    # reference stream is stretched by its reset rows), so compare per-env sequences
    T, B = G["done"].shape
    ref_by_env = [[] for _ in range(B)]
    qi = 0
    for t in range(T):
        for b in np.nonzero(G["done"][t])[0]:
            ref_by_env[b].append((G["mean_power_queue"][qi], G["return_queue"][qi], G["length_queue"][qi]))
            qi += 1
    assert qi == len(G["mean_power_queue"])
    got_by_env = [[] for _ in range(B)]
    # replay our wrapper again, recording which env each entry belongs to
    env2 = _ScriptedVecEnv()
    w2 = RecordEpisodeVals(env2, buffer_length=10000)
    w2.reset()
    n_prev = 0
    for _ in range(env2.n_steps):
        _, _, _, trunc, _ = w2.step(None)
        done_b = np.nonzero(trunc)[0]
        new = list(w2.mean_power_queue)[n_prev:]
        assert len(new) == len(done_b)
        for j, b in enumerate(done_b):
            got_by_env[b].append((w2.mean_power_queue[n_prev + j], w2.return_queue[n_prev + j], w2.length_queue[n_prev + j]))
        n_prev += len(done_b)
    n_cmp = 0
    for b in range(B):
        assert len(got_by_env[b]) >= 3
        for got, ref in zip(got_by_env[b], ref_by_env[b]):
            assert got[2] == ref[2]
            assert got[0] == ref[0], (b, got, ref)                 # bit-identical float64 means
            np.testing.assert_allclose(got[1], ref[1], rtol=1e-13, atol=1e-13)
            n_cmp += 1
    assert n_cmp >= 60 and list(w.mean_power_queue) == list(w2.mean_power_queue)


def test_oracle_accumulators_equal_the_host_wrapper(oracle_lib):
    """The device-style running sums (WG_MET_*) are the wrapper's queues in reduced form."""
    from windgym_amd.config import EnvConfig
    from windgym_amd.envs import RecordEpisodeVals
    from windgym_amd.presets import env1_config
    from windgym_amd.turbine import V80
    d = env1_config()
    d["ActionMethod"] = "yaw"
    B = 7
    cfg = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None", n_envs=B, autoreset=True, n_passthrough=0.5, n_rotor_pts=4)
    orc = oracle_lib.Oracle(cfg)
    orc.reset(seeds=50 + np.arange(B))
    rng = np.random.default_rng(1)

    class _OracleVec:
        as_torch = False
        num_envs = B

        def reset(self, **kw):
            return None, {}

        def step(self, actions):
            obs, rew, tr, _ = orc.step(actions)
            return obs, rew, np.zeros(B, dtype=bool), tr, {"Power agent": orc.info("step_power_agent")}

    w = RecordEpisodeVals(_OracleVec(), buffer_length=100000)
    w.reset()
    for _ in range(400):
        w.step(rng.uniform(-1, 1, size=(B, cfg.n_turb)).astype(np.float32))
    m = orc.metrics()
    n_ep = len(w.mean_power_queue)
    assert n_ep >= 2 * B and m[3] == n_ep and m[7] == 400 * B
    np.testing.assert_allclose(m[2], np.sum(w.mean_power_queue), rtol=1e-12)
    np.testing.assert_allclose(m[0], np.sum(w.return_queue), rtol=1e-9, atol=1e-9)
    assert m[1] == np.sum(w.length_queue)
