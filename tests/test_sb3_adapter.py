"""SB3VecEnv (windgym_amd/envs.py): the stable-baselines3 VecEnv protocol over the batched env, checked on the CPU with a
stand-in for the GPU batch (the adapter itself never touches the device; the real pairing runs in tests/test_gpu_envs.py).
Reference usage: examples/longer_steps_example.py:194-209 (make_vec_env + SubprocVecEnv)."""
import numpy as np

from windgym_amd.envs import SB3VecEnv


class _Box:
    def __init__(self, shape):
        self.shape = shape


class _FakeVec:
    """the part of WindFarmVecEnv the adapter uses"""
    as_torch = False

    def __init__(self, n_envs=6, n_turb=4, obs_dim=8, trunc_every=5):
        self.num_envs, self.n_turb, self.obs_dim, self.trunc_every = n_envs, n_turb, obs_dim, trunc_every
        self.single_observation_space, self.single_action_space = _Box((obs_dim,)), _Box((n_turb,))
        self.t = 0
        self.closed = False
        self.seeded = None

    def _obs(self, k):
        return np.full((self.num_envs, self.obs_dim), float(k), dtype=np.float32)

    def reset(self, **kw):
        self.t = 0
        return self._obs(0), {"Power agent": np.zeros(self.num_envs)}

    def step(self, actions):
        assert np.asarray(actions).shape == (self.num_envs, self.n_turb)
        self.t += 1
        trunc = np.zeros(self.num_envs, dtype=bool)
        if self.t % self.trunc_every == 0:
            trunc[::2] = True
        obs = self._obs(self.t)
        obs[trunc] = -1.0                                   # first observation of the swapped-in episode
        infos = {"final_obs": self._obs(self.t), "Power agent": np.arange(self.num_envs, dtype=np.float64) * self.t}
        return obs, np.ones(self.num_envs, dtype=np.float32), np.zeros_like(trunc), trunc, infos

    def close(self):
        self.closed = True

    def seed(self, seed=None):
        self.seeded = seed
        return [seed] * self.num_envs


def test_sb3_protocol_shapes_and_terminal_observation():
    v = _FakeVec()
    env = SB3VecEnv(v)
    assert env.num_envs == 6 and env.observation_space.shape == (8,) and env.action_space.shape == (4,)
    obs = env.reset()                                       # SB3: observations only, not (obs, info)
    assert isinstance(obs, np.ndarray) and obs.shape == (6, 8)
    n_term = 0
    for k in range(1, 11):
        env.step_async(np.zeros((6, 4), dtype=np.float32))
        obs, rew, dones, infos = env.step_wait()
        assert obs.shape == (6, 8) and rew.shape == (6,) and dones.dtype == bool and len(infos) == 6
        for i in range(6):
            assert infos[i]["TimeLimit.truncated"] == bool(dones[i])
            assert infos[i]["Power agent"] == float(i * k)
            if dones[i]:                                    # same-step reset: obs is the new episode's, the old one's
                n_term += 1                                 # last observation travels in the info dict
                assert (obs[i] == -1.0).all() and (infos[i]["terminal_observation"] == float(k)).all()
            else:
                assert "terminal_observation" not in infos[i]
    assert n_term == 6
    o2, r2, d2, i2 = env.step(np.zeros((6, 4), dtype=np.float32))     # step() = step_async + step_wait
    assert o2.shape == (6, 8) and len(i2) == 6


def test_sb3_housekeeping_methods():
    v = _FakeVec()
    env = SB3VecEnv(v)
    assert env.env_is_wrapped(object) == [False] * 6 and env.env_is_wrapped(object, indices=[1, 2]) == [False, False]
    assert env.get_attr("n_turb") == [4] * 6 and env.get_attr("n_turb", indices=3) == [4]
    assert env.seed(7) == [7] * 6 and v.seeded == 7
    assert env.get_images() == [None] * 6 and env.render() is None
    for call in (lambda: env.set_attr("x", 1), lambda: env.env_method("foo")):
        try:
            call()
            raise AssertionError("expected NotImplementedError")
        except NotImplementedError:
            pass
    env.close()
    assert v.closed


def test_vec_env_subclasses_gymnasium_vectorenv_when_gymnasium_is_installed(monkeypatch):
    """The reference's RecordEpisodeVals extends gymnasium.wrappers.vector.RecordEpisodeStatistics
    (wrappers/recordEpisodeVals.py:8), which only accepts a gymnasium VectorEnv: with gymnasium importable,
    WindFarmVecEnv must be one (same-step autoreset declared in its metadata)."""
    import importlib
    import sys
    import types
    gym = types.ModuleType("gymnasium")
    vec = types.ModuleType("gymnasium.vector")

    class VectorEnv:
        pass

    class AutoresetMode:
        SAME_STEP = "same_step_enum"

    vec.VectorEnv, vec.AutoresetMode = VectorEnv, AutoresetMode
    gym.vector = vec
    monkeypatch.setitem(sys.modules, "gymnasium", gym)
    monkeypatch.setitem(sys.modules, "gymnasium.vector", vec)
    import windgym_amd.envs as envs
    try:
        mod = importlib.reload(envs)
        assert issubclass(mod.WindFarmVecEnv, VectorEnv)
        assert mod._gym_autoreset_mode() == "same_step_enum"
    finally:
        monkeypatch.undo()
        importlib.reload(envs)
    assert envs.WindFarmVecEnv.__mro__[1] is object
