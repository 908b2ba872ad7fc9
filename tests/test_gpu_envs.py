"""GPU tests of the host classes: the reference's own invariant tests (tests/test_basics.py) replayed against
WindFarmEnv / FarmEval / WindFarmEnvMulti / WindFarmVecEnv, and the known-answer yaw trajectory of
test_fast_eval (tests/test_basics.py:415-471)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def wg():
    import torch
    assert torch.cuda.is_available()
    import windgym_amd
    return windgym_amd


def _yaml(tmp_path, d, name="cfg.yaml"):
    import yaml
    p = tmp_path / name
    p.write_text(yaml.safe_dump(d))
    return str(p)


@pytest.fixture(params=["two_turb", "env1"])
def env(request, wg, tmp_path):
    from windgym_amd import presets
    d = presets.two_turb_config() if request.param == "two_turb" else presets.env1_config()
    e = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=2, yaml_path=_yaml(tmp_path, d), turbtype="None", seed=3)
    yield e
    e.close()


def test_environment_initialization(env):            # tests/test_basics.py:100-113
    assert isinstance(env.observation_space.sample(), np.ndarray)
    assert isinstance(env.action_space.sample(), np.ndarray)
    assert env.n_turb > 0 and hasattr(env, "turbine")


def test_environment_reset(env):                     # :116-137
    obs, info = env.reset()
    assert isinstance(obs, np.ndarray) and obs.dtype == np.float32
    assert obs.shape == env.observation_space.shape
    assert not np.any(np.isnan(obs))
    for key in ["yaw angles agent", "Wind speed Global", "Wind direction Global", "Power agent"]:
        assert key in info


def test_environment_step_and_limits(env):           # :140-209
    env.reset()
    for _ in range(3):
        obs, reward, terminated, truncated, info = env.step(np.zeros(env.action_space.shape))
        assert obs.shape == env.observation_space.shape
        assert isinstance(reward, float) and not np.isnan(reward)
        assert isinstance(terminated, bool) and isinstance(truncated, bool) and isinstance(info, dict)
        assert info["Power agent"] >= 0
    _, _, _, _, info = env.step(np.ones(env.action_space.shape))
    assert np.all(info["yaw angles agent"] >= env.yaw_min) and np.all(info["yaw angles agent"] <= env.yaw_max)
    env.action_penalty, env.action_penalty_type = 1.0, "Change"
    assert env._action_penalty() >= 0
    env.action_penalty_type = "Total"
    assert env._action_penalty() >= 0
    assert env._get_num_raw_features() <= env.obs_var


def test_wind_conditions_and_truncation(env):        # :232-246 + truncation semantics (:1003-1025)
    _, info = env.reset()
    assert env.ws_min <= info["Wind speed Global"] <= env.ws_max
    assert env.wd_min <= info["Wind direction Global"] <= env.wd_max
    assert len(info["Wind speed at turbines"]) == env.n_turb
    n = 0
    while True:
        _, _, _, truncated, _ = env.step(env.action_space.sample())
        n += 1
        if truncated:
            break
    assert n == env.time_max + 1
    with pytest.raises(RuntimeError):
        env.step(env.action_space.sample())
    env.reset()
    env.step(env.action_space.sample())


def test_fast_eval_known_answer(wg, tmp_path):       # tests/test_basics.py:415-471
    from windgym_amd import presets
    env = wg.FarmEval(turbine=wg.V80(), yaml_path=_yaml(tmp_path, presets.env1_config()), turbtype="None",
                      yaw_init="Zeros", seed=1)
    env.set_wind_vals(ws=10, ti=0.07, wd=270)
    obs, info = env.reset()
    assert np.allclose(env.fs.windTurbines.yaw, 0.0)
    assert env.ws == 10 and env.ti == 0.07 and env.wd == 270
    yaw_goal = np.array([-10.0, 20.0, 0.0, 0.0])
    action = ((yaw_goal - env.yaw_min) / (env.yaw_max - env.yaw_min) * 2 - 1).astype(np.float32)  # BaseAgent.scale_yaw
    for i in range(1, 50):
        obs, reward, terminated, truncated, info = env.step(action)
        assert not truncated
    assert np.allclose(env.fs.windTurbines.yaw, yaw_goal, atol=1e-4)
    assert env.time_max == 9999999
    # the two downstream turbines of the 2x2 farm are waked, the upstream pair is not
    u = env.fs.windTurbines.rotor_avg_windspeed[:, 0]
    assert np.allclose(u[[0, 2]], 10.0, atol=1e-4) and np.all(u[[1, 3]] < 10.0)
    env.close()


def test_multi_agent_facade(wg, tmp_path):
    from windgym_amd import presets
    d = presets.env1_config()
    d["farm"].update(nx=3, ny=3)
    env = wg.WindFarmEnvMulti(turbine=wg.V80(), n_passthrough=2, yaml_path=_yaml(tmp_path, d), turbtype="None", seed=51)
    obs, infos = env.reset()
    assert env.possible_agents == [f"turbine_{i}" for i in range(9)] and set(obs) == set(env.possible_agents)
    assert env.observation_space("turbine_0").shape == (env.obs_var,)
    assert env.action_space("turbine_0").shape == (1,)
    steps = 0
    while env.agents:
        acts = {a: env.action_space(a).sample() for a in env.agents}
        obs, rewards, terms, truncs, infos = env.step(acts)
        steps += 1
        assert all(v.shape == (env.obs_len,) and v.dtype == np.float32 for v in obs.values())
        assert len(set(rewards.values())) == 1 and not any(terms.values())
    # timestep advances twice per step (WindEnvMulti.py:219): the episode lasts ceil(time_max / 2) + 1 steps
    assert steps == (env.time_max + 1) // 2 + 1
    env.close()


def test_vec_env_autoreset_and_episode_stats(wg, tmp_path):
    import torch
    from windgym_amd import presets
    from windgym_amd.envs import RecordEpisodeVals
    d = presets.env1_config()
    d["ActionMethod"] = "yaw"
    venv = wg.WindFarmVecEnv(wg.V80(), 64, yaml_path=_yaml(tmp_path, d), turbtype="None", n_passthrough=1, seed=7,
                             as_torch=True)
    rec = RecordEpisodeVals(venv, buffer_length=100000)
    obs, infos = rec.reset()
    assert obs.shape == (64, venv.single_observation_space.shape[0]) and obs.is_cuda
    n_done = 0
    for i in range(400):
        a = torch.rand((64, venv.n_turb), device="cuda") * 2 - 1
        obs, rew, term, trunc, infos = rec.step(a)
        assert not term.any() and torch.isfinite(obs).all() and torch.isfinite(rew).all()
        assert (obs.abs() <= 1).all()
        n_done += int(trunc.sum())
        assert (infos["Power agent"] >= 0).all()
    venv.batch.check()
    assert n_done >= 64 and len(rec.mean_power_queue) == n_done
    m = venv.metrics()
    assert m["n_episodes"] == n_done and m["n_steps"] == 400 * 64
    # the device accumulators ARE the wrapper's queues in reduced form (recordEpisodeVals.py:43-56; the wrapper's own
    # semantics are pinned against the reference's file in tests/test_record_episode_vals.py): infos["Power agent"] of a
    # truncated env is the terminal step's power, so both sides average the same steps — fp32 running sums on the
    # device, float64 on the host
    np.testing.assert_allclose(m["ep_mean_power_sum"], np.sum(rec.mean_power_queue), rtol=2e-6)
    np.testing.assert_allclose(m["ep_return_sum"], np.sum(rec.return_queue), rtol=1e-4, atol=1e-3)
    assert m["ep_length_sum"] == np.sum(rec.length_queue)
    # stable-baselines3's VecEnv protocol: reset() returns the observations only, per-env spaces,
    # step_async / step_wait with a list of per-env info dicts and terminal_observation on same-step resets
    sb3 = venv.as_sb3()
    assert sb3.num_envs == 64 and sb3.observation_space.shape == (venv.batch.obs_dim,)
    assert sb3.action_space.shape == (venv.n_turb,)
    o = sb3.reset()
    assert isinstance(o, np.ndarray) and o.shape == (64, venv.batch.obs_dim)
    n_term = 0
    for _ in range(200):
        sb3.step_async(np.zeros((64, venv.n_turb), dtype=np.float32))
        o, r, dones, inf = sb3.step_wait()
        assert o.shape == (64, venv.batch.obs_dim) and r.shape == (64,) and dones.dtype == bool
        assert len(inf) == 64 and "Power agent" in inf[0]
        for i in np.nonzero(dones)[0]:
            assert inf[i]["TimeLimit.truncated"] and inf[i]["terminal_observation"].shape == (venv.batch.obs_dim,)
            n_term += 1
        assert all("terminal_observation" not in inf[i] for i in np.nonzero(~dones)[0])
    assert n_term > 0
    assert sb3.env_is_wrapped(object) == [False] * 64 and len(sb3.get_attr("n_turb")) == 64
    sb3.close()


def test_mann_box_generators_agree_statistically(wg):
    """hipFFT (torch) generator vs the numpy reference implementation of the same algorithm."""
    import torch
    from windgym_amd.mann import generate_mann_box, generate_mann_box_torch
    a = generate_mann_box((256, 64, 32), (3.0, 3.0, 3.0), seed=5)
    b = generate_mann_box_torch((256, 64, 32), (3.0, 3.0, 3.0), seed=5).cpu().numpy()
    assert a.shape == b.shape == (3, 256, 64, 32)
    for arr in (a, b):
        assert abs(arr[0].std() - 1.0) < 1e-3
        assert 0.6 < arr[1].std() < 0.95 and 0.5 < arr[2].std() < 0.95
        assert np.mean(arr[0] * arr[2]) < -0.1                     # shear: negative uw covariance
    assert abs(a[1].std() - b[1].std()) < 0.08 and abs(a[2].std() - b[2].std()) < 0.08
    ac = lambda u, k: np.mean(u[:-k] * u[k:])                      # noqa: E731
    assert abs(ac(a[0], 10) - ac(b[0], 10)) < 0.15


def test_mannfixed_env_with_reference_box_size(wg, tmp_path):
    """turbtype='MannFixed' end to end: the 2048x512x64 box (0.8 GB) is generated on the GPU and the env steps."""
    from windgym_amd import presets
    d = presets.env1_config()
    env = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=1, yaml_path=_yaml(tmp_path, d), turbtype="MannFixed", seed=2)
    obs, info = env.reset()
    assert np.isfinite(obs).all() and np.std(info["Wind direction at turbines"]) > 0
    u0 = env.fs.windTurbines.rotor_avg_windspeed.copy()
    for _ in range(5):
        obs, r, term, trunc, info = env.step(env.action_space.sample())
    assert np.isfinite(obs).all() and np.isfinite(r)
    assert np.abs(env.fs.windTurbines.rotor_avg_windspeed - u0).max() > 1e-3    # unsteady inflow
    env.close()


def test_measured_info_entries_are_unscaled_windows(wg, tmp_path):
    """'... measured' info entries = unscaled MesClass windows; scaling them reproduces the observation."""
    from windgym_amd import presets
    d = presets.env1_config()
    d["mes_level"].update(turb_wd=True, farm_ws=True, farm_wd=True)
    d["wd_mes"].update(wd_rolling_mean=True, wd_current=True)
    env = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=2, yaml_path=_yaml(tmp_path, d), turbtype="None", seed=4)
    obs, info = env.reset()
    for _ in range(3):
        obs, _, _, _, info = env.step(env.action_space.sample())
    ws_m = info["Wind speed at turbines measured"]
    assert ws_m.shape == (env.n_turb,) and np.all((ws_m > 2) & (ws_m < 25))
    lay = env.cfg.obs_layout()
    o, n = lay["turb"]["ws"]
    scaled = np.array([obs[t * lay["turb_block"] + o] for t in range(env.n_turb)])
    np.testing.assert_allclose(2 * (ws_m - 2.0) / 23.0 - 1, scaled, atol=1e-5)
    assert info["yaw angles measured"].shape == (env.n_turb,)
    np.testing.assert_allclose(info["yaw angles measured"], np.mean([info["yaw angles agent"]], axis=0), atol=5.0)
    assert info["Wind direction at turbines measured"].shape == (2 * env.n_turb,)
    assert info["Wind speed at farm measured"].shape == (1,)
    assert abs(info["Wind speed at farm measured"][0] - ws_m.mean()) < 0.5
    env.close()


def test_invalid_configurations_raise_like_the_reference(wg, tmp_path):
    from windgym_amd import presets
    base = presets.env1_config()
    bad = dict(base, ActionMethod="absolute")
    with pytest.raises(NotImplementedError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, bad), turbtype="None")
    bad = dict(base, ActionMethod="sideways")
    with pytest.raises(ValueError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, bad), turbtype="None")
    bad = dict(base, power_def=dict(Power_reward="Power_diff", Power_avg=10, Power_scaling=1.0))
    with pytest.raises(ValueError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, bad), turbtype="None")
    bad = dict(base, power_def=dict(Power_reward="Nope", Power_avg=10, Power_scaling=1.0))
    with pytest.raises(ValueError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, bad), turbtype="None")
    bad = dict(base, Track_power=True)
    with pytest.raises(NotImplementedError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, bad), turbtype="None")
    with pytest.raises(ValueError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, base), turbtype="Gusty")
    with pytest.raises(ValueError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, base), turbtype="None", dt_sim=2, dt_env=3)
    with pytest.raises(ValueError):          # C-ABI level validation (n_particles must be a multiple of 4)
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, base), turbtype="None", n_particles=30)
    with pytest.raises(NotImplementedError):
        wg.WindFarmEnv(turbine=wg.V80(), yaml_path=_yaml(tmp_path, base), turbtype="None", HTC_path="x.htc")


def test_batched_eval_sweep_matches_single_env_rollouts(wg, tmp_path):
    """f3: one batched rollout over (ws x wd x TI) == the reference-style one-condition-at-a-time evaluation
    (tests/test_basics.py:415-471 scenario: ConstantAgent [-10, 20, 0, 0], FarmEval, turbtype None)."""
    from windgym_amd import presets
    from windgym_amd.agents import ConstantAgent
    from windgym_amd.evaluate import eval_sweep
    ypath = _yaml(tmp_path, presets.env1_config())
    yaw_goal = np.array([-10.0, 20.0, 0.0, 0.0])
    wss, wds, tis = (8.0, 10.0), (265.0, 270.0, 275.0), (0.07,)
    ds = eval_sweep(wg.V80(), ypath, ConstantAgent(yaw_goal), winddirs=wds, windspeeds=wss, turbintensities=tis,
                    t_sim=40, turbtype="None")
    d = ds["data"] if isinstance(ds, dict) else {k: ds[k].values for k in ds.data_vars}
    assert d["yaw_a"].shape == (40, 4, 2, 3, 1, 1, 1) and d["powerF_a"].shape == (40, 2, 3, 1, 1, 1)
    assert np.allclose(d["yaw_a"][0], 0.0) and np.allclose(d["yaw_a"][-1][:, 0, 0, 0, 0, 0], yaw_goal, atol=1e-4)
    assert np.allclose(d["pct_inc"][0], 0.0, atol=1e-3)          # identical farms before the first action
    # one condition re-run through the single-env facade gives the same trajectory
    env = wg.FarmEval(turbine=wg.V80(), yaml_path=ypath, turbtype="None", yaw_init="Zeros", seed=1, Baseline_comp=True)
    env.set_wind_vals(ws=10.0, ti=0.07, wd=275.0)
    env.reset(seed=1)
    agent = ConstantAgent(yaw_goal)
    agent.yaw_max, agent.yaw_min = env.yaw_max, env.yaw_min
    for i in range(1, 40):
        env.step(agent.predict(None)[0])
    np.testing.assert_allclose(env.fs.windTurbines.power(), d["powerT_a"][-1][:, 1, 2, 0, 0, 0], rtol=1e-5)
    np.testing.assert_allclose(env.fs_baseline.windTurbines.power(), d["powerT_b"][-1][:, 1, 2, 0, 0, 0], rtol=1e-5)
    env.close()


def test_greedy_agent_reproduces_the_baseline_controller(wg, tmp_path):
    """GreedyAgent('local') through the 'wind' action method == the baseline farm driven by the local controller:
    the Baseline reward stays ~0 and the yaws of both farms agree."""
    from windgym_amd import presets
    from windgym_amd.agents import GreedyAgent
    d = presets.env1_config()
    venv = wg.WindFarmVecEnv(wg.V80(), 8, yaml_path=_yaml(tmp_path, d), turbtype="Random", n_passthrough=2, seed=3)
    agent = GreedyAgent(type="local", env=venv, yaw_max=venv.cfg.yaw_max, yaw_min=venv.cfg.yaw_min)
    obs, _ = venv.reset()
    for _ in range(40):
        a, _ = agent.predict(obs)
        obs, rew, term, trunc, infos = venv.step(a)
    ya, yb = infos["yaw angles agent"], infos["yaw angles base"]
    assert np.abs(ya - yb).max() < 0.3           # both follow the local wind direction, 1 deg/step limited
    assert np.abs(rew).max() < 0.02
    venv.close()


def test_render_and_flow_field_view(wg, tmp_path):
    """render_mode="rgb_array" / plot_frame / fs.get_windspeed (Wind_Farm_Env.py:464-476, :1036-1103)."""
    from windgym_amd import presets
    env = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=2, yaml_path=_yaml(tmp_path, presets.env1_config()),
                         turbtype="None", seed=5, render_mode="rgb_array", Baseline_comp=True)
    env.reset(seed=5)
    for _ in range(5):
        env.step(env.action_space.sample())
    env.init_render()
    assert env.a.shape == (250,) and env.b.shape == (250,)
    uvw = env.fs.get_windspeed(env.view, include_wakes=True)
    assert uvw.shape == (3, 250, 250) and np.isfinite(uvw).all()
    ws = float(env.ws)
    assert uvw[0].max() <= ws + 1e-3 and uvw[0].min() < ws - 1.0          # wakes, no speed-up in model M0
    # the field at a turbine's hub is consistent with what the turbine sees (rotor average of the same field)
    x_t, y_t = env.fs.windTurbines.positions_xyz[:2]
    t_last = int(np.argmax(x_t))
    u_hub = env.get_windspeed(x=[x_t[t_last] - 1.0], y=[y_t[t_last]])[0, 0, 0]      # just upstream of its own wake
    u_rot = env.fs.windTurbines.rotor_avg_windspeed[t_last, 0]
    assert abs(u_hub - u_rot) < 0.35 * max(ws - u_rot, 0.5) + 0.05
    img = env.render()
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3 and img.std() > 5
    img_b = env.plot_frame(baseline=True)
    assert img_b.shape == img.shape
    env.close()


def test_set_windconditions_with_site(wg, tmp_path):      # tests/test_basics.py:369-407
    """sample_site: wd / ws from the site's wind rose, clipped to the env's ranges; TI stays uniform."""
    from windgym_amd import presets
    from windgym_amd.site import hornsrev1_site
    env = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=1, yaml_path=_yaml(tmp_path, presets.env1_config()),
                         turbtype="None", seed=7, sample_site=hornsrev1_site())
    samples = []
    for _ in range(10):
        env._set_windconditions()
        samples.append((env.ws, env.wd, env.ti))
    for ws, wd, ti in samples:
        assert env.ws_min <= ws <= env.ws_max and env.wd_min <= wd <= env.wd_max and env.TI_min <= ti <= env.TI_max
    assert len({s[0] for s in samples}) > 1 and len({s[2] for s in samples}) > 1
    # the episodes themselves use the site's draw: integer directions (1-degree wind rose), clipped speeds
    seen = []
    for i in range(6):
        env.reset(seed=100 + i)
        assert env.ws_min <= env.ws <= env.ws_max and env.wd_min <= env.wd <= env.wd_max
        assert float(env.wd).is_integer()
        assert env.TI_min <= env.ti <= env.TI_max
        seen.append((env.ws, env.wd, env.ti))
    assert len({s[0] for s in seen}) > 1 and len({s[2] for s in seen}) > 1
    env.reset(seed=100)                                   # reproducible per seed
    assert (env.ws, env.wd, env.ti) == seen[0]
    env.close()

    # batched: every new episode of every env gets a fresh draw from the device-resident table
    d = presets.two_turb_config()
    d["wind"].update(ws_min=6, ws_max=14, wd_min=250, wd_max=290)
    venv = wg.WindFarmVecEnv(wg.V80(), 64, yaml_path=_yaml(tmp_path, d, "v.yaml"), turbtype="None", seed=1,
                             n_passthrough=1, sample_site=hornsrev1_site())
    venv.reset(seed=1)
    first = np.stack([venv.infos()["Wind speed Global"], venv.infos()["Wind direction Global"]], axis=1).copy()
    assert (first[:, 0] >= 6).all() and (first[:, 0] <= 14).all() and (first[:, 1] >= 250).all() and (first[:, 1] <= 290).all()
    assert np.allclose(first[:, 1], np.round(first[:, 1])) and len(np.unique(first[:, 0])) > 20
    n_tr = 0
    for _ in range(400):
        _, _, _, tr, _ = venv.step(np.zeros((64, venv.n_turb), dtype=np.float32))
        n_tr += int(np.sum(tr))
    assert n_tr >= 64
    later = np.stack([venv.infos()["Wind speed Global"], venv.infos()["Wind direction Global"]], axis=1)
    assert (later[:, 0] >= 6).all() and (later[:, 0] <= 14).all() and (later[:, 1] >= 250).all() and (later[:, 1] <= 290).all()
    assert np.allclose(later[:, 1], np.round(later[:, 1]))
    assert (np.abs(later[:, 0] - first[:, 0]) > 1e-6).mean() > 0.8     # new episodes, new draws
    venv.batch.check()
    venv.close()


def test_mannload_reads_the_turbbox_file_or_falls_back(wg, tmp_path, capsys):
    """turbtype "MannLoad": TurbBox = a box file / a directory of TF_* files (Wind_Farm_Env.py:197-213, :611-618);
    without files the reference switches to generated turbulence."""
    from windgym_amd import presets
    from windgym_amd.mann import generate_mann_box, save_box
    d = tmp_path / "boxes"
    d.mkdir()
    box = generate_mann_box((128, 32, 16), (3.0, 3.0, 3.0), seed=11)
    save_box(str(d / "TF_test.npz"), box, (3.0, 3.0, 3.0))
    y = _yaml(tmp_path, presets.env1_config())
    env = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=1, yaml_path=y, turbtype="MannLoad", TurbBox=str(d), seed=2)
    ref = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=1, yaml_path=y, turbtype="MannLoad", seed=2,
                         turbulence_box=(box, (3.0, 3.0, 3.0)))
    for _ in range(25):
        a = env.action_space.sample()
        o1, r1, *_ = env.step(a)
        o2, r2, *_ = ref.step(a)
        # the file round-trips to the same field (load_box renormalises to unit std: one float rounding of the box values)
        np.testing.assert_allclose(o1, o2, rtol=0, atol=1e-6)
    v = env.fs.windTurbines.rotor_avg_windspeed[:, 1]
    assert np.abs(v).max() > 1e-3                             # turbulent inflow is on
    env.close(), ref.close()
    capsys.readouterr()
    env = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=1, yaml_path=y, turbtype="MannLoad",
                         TurbBox=str(tmp_path / "nowhere"), seed=2)
    assert "switch to generated turbulence" in capsys.readouterr().out
    env.step(env.action_space.sample())
    env.close()


def test_agent_eval_facade(wg, tmp_path):
    """AgentEval(env, model, name, t_sim).set_conditions / eval_multiple / save_performance / load_performance
    (AgentEval.py:478-715; tests/test_basics.py:327-365 drives it this way)."""
    from windgym_amd import presets
    env = wg.FarmEval(turbine=wg.V80(), yaml_path=_yaml(tmp_path, presets.env1_config()), turbtype="None",
                      yaw_init="Zeros", seed=1)
    tester = wg.AgentEval(env=env, model=wg.ConstantAgent(yaw_angles=[0, 0, 0, 0]), name=str(tmp_path / "const"), t_sim=12)
    tester.set_conditions(winddirs=[260, 270], windspeeds=[10], turbintensities=[0.07], turbboxes=["Default"])
    ds = tester.eval_multiple(save_figs=False, debug=False)
    data = ds["data"] if isinstance(ds, dict) else {k: ds[k].values for k in ds.data_vars}
    assert data["powerF_a"].shape == (12, 1, 2, 1, 1, 1) and data["powerT_a"].shape == (12, 4, 1, 2, 1, 1, 1)
    assert (data["powerF_a"] > 0).all() and "pct_inc" in data
    path = tester.save_performance()
    other = wg.AgentEval(env=env, model=None, name="x")
    ds2 = other.load_performance(path)
    np.testing.assert_array_equal(ds2["data"]["powerF_a"], data["powerF_a"])
    assert list(ds2["coords"]["wd"]) == [260.0, 270.0]
    env.close()
