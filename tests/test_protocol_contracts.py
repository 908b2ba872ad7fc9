"""The host facades against the WRITTEN contracts of the libraries the reference's users plug them into.
gymnasium, stable-baselines3 and pettingzoo are not installed in the build image, so their own checkers cannot run; what they
check is restated here from their documented contracts:

  * stable-baselines3 `VecEnv` (stable_baselines3/common/vec_env/base_vec_env.py): the abstract methods a subclass must define
    and the signatures `VecEnvWrapper`s call them with (CPU: the adapter never touches the device);
  * gymnasium `utils.env_checker.check_env` (what the reference's tests/test_basics.py:410-412 runs): spaces, `reset(seed=)` ->
    (obs in space, info dict) and its determinism, `step` -> (obs in space, finite float reward, bool, bool, dict) (GPU);
  * pettingzoo `ParallelEnv`: `possible_agents` / `agents`, `reset -> (obs, infos)`, `step(dict) -> 5 dicts keyed by agent`,
    `observation_space(agent)` / `action_space(agent)` stable across calls (GPU).
"""
import inspect

import numpy as np
import pytest

# ---------------------------------------------------------------------------------------------------------------------
# stable-baselines3 VecEnv: abstract methods and the call signatures of the base class
# ---------------------------------------------------------------------------------------------------------------------
SB3_ABSTRACT = {          # name -> parameter names after self (base_vec_env.VecEnv)
    "reset": [],
    "step_async": ["actions"],
    "step_wait": [],
    "close": [],
    "get_attr": ["attr_name", "indices"],
    "set_attr": ["attr_name", "value", "indices"],
    "env_method": ["method_name", "method_args", "indices", "method_kwargs"],
    "env_is_wrapped": ["wrapper_class", "indices"],
}
SB3_CONCRETE = ["step", "seed", "get_images", "render"]        # the base class implements them; wrappers call them


def test_sb3_vecenv_abstract_methods_and_signatures():
    from windgym_amd.envs import SB3VecEnv
    for name, params in SB3_ABSTRACT.items():
        fn = getattr(SB3VecEnv, name, None)
        assert callable(fn), f"SB3VecEnv lacks the abstract method {name}"
        got = [p for p in inspect.signature(fn).parameters if p != "self"]
        assert got == params, (name, got, params)
    for name in SB3_CONCRETE:
        assert callable(getattr(SB3VecEnv, name, None)), name
    sig = inspect.signature(SB3VecEnv.get_attr)
    assert sig.parameters["indices"].default is None
    kinds = {n: p.kind for n, p in inspect.signature(SB3VecEnv.env_method).parameters.items()}
    assert kinds["method_args"] == inspect.Parameter.VAR_POSITIONAL and kinds["method_kwargs"] == inspect.Parameter.VAR_KEYWORD
    assert kinds["indices"] == inspect.Parameter.KEYWORD_ONLY


def test_sb3_vecenv_attributes_wrappers_read():
    """VecEnvWrapper / VecMonitor / evaluate_policy read these attributes of the wrapped VecEnv"""
    from tests.test_sb3_adapter import _FakeVec
    from windgym_amd.envs import SB3VecEnv
    env = SB3VecEnv(_FakeVec())
    for attr in ("num_envs", "observation_space", "action_space", "render_mode"):
        assert hasattr(env, attr), attr
    assert isinstance(env.num_envs, int)
    obs = env.reset()
    assert obs.shape[0] == env.num_envs
    # step_wait returns (obs, rewards, dones, infos): ndarray, ndarray, ndarray of bool, LIST of dicts (one per env)
    env.step_async(np.zeros((env.num_envs,) + env.action_space.shape, dtype=np.float32))
    o, r, d, infos = env.step_wait()
    assert isinstance(o, np.ndarray) and isinstance(r, np.ndarray) and d.dtype == bool
    assert isinstance(infos, (list, tuple)) and len(infos) == env.num_envs and all(isinstance(i, dict) for i in infos)


# ---------------------------------------------------------------------------------------------------------------------
# gymnasium check_env / pettingzoo ParallelEnv, restated (need the device: the facades drive the HIP batch)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def wg():
    import torch
    import windgym_amd
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return windgym_amd


def _yaml(tmp_path, d):
    import yaml
    p = tmp_path / "cfg.yaml"
    p.write_text(yaml.safe_dump(d))
    return str(p)


@pytest.mark.gpu
def test_check_env_contract_restated(wg, tmp_path):
    from windgym_amd import presets
    from windgym_amd.turbine import V80
    env = wg.WindFarmEnv(V80(), yaml_path=_yaml(tmp_path, presets.env1_config()), turbtype="None", seed=3, n_passthrough=0.5)
    # check_env: the spaces exist, are Boxes of the declared dtype, and sample() lies inside them
    for sp in (env.observation_space, env.action_space):
        assert sp.dtype == np.float32 and sp.contains(sp.sample())
        assert np.all(sp.low == -1.0) and np.all(sp.high == 1.0)
    # check_reset_seed / check_reset_return_type: (obs, info), obs in the space, same seed -> same observation
    o1, i1 = env.reset(seed=123)
    assert isinstance(i1, dict) and isinstance(o1, np.ndarray) and o1.dtype == np.float32 and env.observation_space.contains(o1)
    o2, _ = env.reset(seed=123)
    np.testing.assert_array_equal(o1, o2)
    o3, _ = env.reset(seed=124)
    assert not np.array_equal(o1, o3)
    # check_step_return_type (passive checker): 5-tuple, obs in space, finite real reward, bool flags, dict info
    truncated = False
    n = 0
    while not truncated and n < 2000:
        out = env.step(env.action_space.sample())
        assert isinstance(out, tuple) and len(out) == 5
        obs, reward, terminated, truncated, info = out
        assert isinstance(obs, np.ndarray) and obs.dtype == np.float32 and env.observation_space.contains(obs)
        assert isinstance(reward, (float, int, np.floating)) and np.isfinite(reward)
        assert isinstance(terminated, (bool, np.bool_)) and isinstance(truncated, (bool, np.bool_)) and not terminated
        assert isinstance(info, dict)
        n += 1
    assert truncated                                   # the time limit ends the episode, never `terminated` (:1003-1025)
    # a reset after truncation starts a fresh episode
    o4, _ = env.reset(seed=5)
    assert env.observation_space.contains(o4)
    env.close()


@pytest.mark.gpu
def test_parallel_env_contract_restated(wg, tmp_path):
    from windgym_amd import presets
    from windgym_amd.turbine import V80
    env = wg.WindFarmEnvMulti(V80(), yaml_path=_yaml(tmp_path, presets.env1_config()), turbtype="None", seed=4, n_passthrough=0.5)
    assert env.possible_agents == [f"turbine_{i}" for i in range(env.n_turb)]           # WindEnvMulti.py:72-77
    obs, infos = env.reset(seed=9)
    assert set(obs) == set(env.agents) == set(env.possible_agents) and set(infos) == set(env.agents)
    for a in env.agents:
        assert env.observation_space(a) is env.observation_space(a) or env.observation_space(a).shape == env.observation_space(a).shape
        assert env.action_space(a).shape == (1,)
        assert obs[a].dtype == np.float32 and obs[a].shape[0] <= env.observation_space(a).shape[0]      # (Appendix B7: shorter than declared)
    done = False
    n = 0
    while not done and n < 2000:
        acts = {a: env.action_space(a).sample() for a in env.agents}
        out = env.step(acts)
        assert isinstance(out, tuple) and len(out) == 5
        o, r, term, trunc, inf = out
        for dct in out:
            assert isinstance(dct, dict) and set(dct) == set(env.possible_agents)
        assert len({float(v) for v in r.values()}) == 1                                  # one shared scalar reward (:205-212)
        assert not any(term.values())
        done = all(trunc.values())
        assert any(trunc.values()) == done                                               # every agent truncates together
        n += 1
    assert done
    env.close()
