"""f4: steady-state limit of model M0 (pinned against the oracle's converged dynamic simulation) and the
serial-refine yaw optimiser built on it (the reference's own test: tests/test_pywake_agent.py:11-45)."""
import numpy as np
import pytest

from windgym_amd.config import EnvConfig
from windgym_amd.presets import env1_config
from windgym_amd.steady import SteadyStateYawAgent, steady_state_power, yaw_optimizer_srf
from windgym_amd.turbine import V80


def test_power_optimization_like_the_reference():
    x_pos, y_pos = [0, 500], [0, 0]
    agent = SteadyStateYawAgent(x_pos=x_pos, y_pos=y_pos, wind_speed=6, wind_dir=270, TI=0.02)
    nominal = agent.power([30, 0])
    agent.optimize()
    assert agent.power(agent.optimized_yaws) >= nominal
    assert agent.power(agent.optimized_yaws) > agent.power([0, 0])          # wake steering pays at 6.25 D, TI 2 %
    assert abs(agent.optimized_yaws[0]) > 5 and abs(agent.optimized_yaws[1]) < 1.0
    a, _ = agent.predict(None)
    assert a.shape == (2,) and np.all(np.abs(a) <= 1)


def test_steady_model_matches_the_converged_dynamic_oracle(oracle_lib):
    """Fixed yaws held for several flow-through times: the dynamic oracle converges to the steady model
    (the deflection integral is continuous here and a dt-sum there: 1 % tolerance on power)."""
    d = env1_config()
    d.update(yaw_init="Zeros", ActionMethod="wind")
    d["wind"].update(ws_min=8.0, ws_max=8.0, TI_min=0.06, TI_max=0.06, wd_min=268.0, wd_max=268.0)
    d["farm"].update(nx=3, ny=2)
    d["power_def"]["Power_reward"] = "Power_avg"
    cfg = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None", never_truncate=True, n_rotor_pts=16)
    o = oracle_lib.Oracle(cfg)
    o.reset(seeds=[0])
    goal = np.array([20.0, -15.0, 10.0, 0.0, 25.0, -5.0])
    a = ((goal + 45.0) / 90.0 * 2 - 1).astype(np.float32)[None]
    for _ in range(400):
        o.step(a)
    p_dyn = o.info("power_turb_agent")[0]
    assert np.allclose(o.info("yaw_agent")[0], goal, atol=1e-4)
    p_st = steady_state_power(cfg.x_pos, cfg.y_pos, 8.0, 268.0, 0.06, goal).numpy()
    np.testing.assert_allclose(p_st, p_dyn, rtol=1e-2)


def test_batched_optimizer_over_conditions():
    x, y = np.meshgrid(np.linspace(0, 1280, 3), np.linspace(0, 640, 2))
    x, y = x.ravel(), y.ravel()
    ws = np.array([7.0, 9.0, 9.0])
    wd = np.array([270.0, 270.0, 250.0])
    ti = np.array([0.04, 0.04, 0.08])
    yaw = yaw_optimizer_srf(x, y, ws, wd, ti, refine_pass_n=4, yaw_n=5)
    assert yaw.shape == (3, 6) and np.all(np.abs(yaw) <= 30.0)
    p_opt = steady_state_power(x, y, ws, wd, ti, yaw).sum(-1).numpy()
    p_zero = steady_state_power(x, y, ws, wd, ti, np.zeros((3, 6))).sum(-1).numpy()
    assert np.all(p_opt >= p_zero - 1e-6)
    assert p_opt[0] > p_zero[0] * 1.01            # aligned rows: steering gains power
