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


# ---- the reference agent's own wake model (Blondel & Cathelain 2020 + Jimenez), restated from the publications ----
def test_super_gaussian_conserves_momentum():
    """The centre-line deficit of the super-Gaussian wake is DERIVED from mass + momentum conservation: for every order
    n and width sigma the momentum-deficit flux through a cross-section must equal the thrust,
    int 2 pi r U (U_inf - U) dr = ct/2 U_inf^2 pi D^2 / 4.  Checked by quadrature — independent of any recalled
    constant."""
    import torch
    from windgym_amd.steady import blondel_centre_deficit
    import math
    r = torch.linspace(0.0, 6.0, 200001, dtype=torch.float64)            # r / D
    checked = 0
    for n in (2.0, 2.41, 3.0, 4.2, 5.5):
        for sigma in (0.28, 0.35, 0.6):
            for ct in (0.3, 0.8):
                if 2 ** (4 / n - 2) - n * ct / (16 * math.gamma(2 / n) * sigma ** (4 / n)) < 0:
                    continue            # narrower than momentum theory allows for this thrust: no real solution
                checked += 1
                C = blondel_centre_deficit(torch.tensor(ct, dtype=torch.float64), torch.tensor(sigma, dtype=torch.float64),
                                           torch.tensor(n, dtype=torch.float64))
                f = C * torch.exp(-r ** n / (2 * sigma ** 2))
                flux = torch.trapezoid(2 * np.pi * r * (1 - f) * f, r)       # in units of U_inf^2 D^2
                assert abs(float(flux) - ct / 2 * np.pi / 4) < 2e-4 * ct, (n, sigma, ct, float(flux))
    assert checked >= 25


def test_blondel_jimenez_single_wake_by_hand():
    """Two V80 in line, 7 D apart, 8 m/s, TI 6 %: the batched model against the equations evaluated by hand."""
    import math
    from windgym_amd.steady import BC_A_F, BC_A_S, BC_B_F, BC_B_S, BC_C_F, BC_C_S, JIMENEZ_BETA, blondel_jimenez_power
    t = V80()
    D, ws, ti, xd = 80.0, 8.0, 0.06, 7.0
    ct = float(np.interp(ws, t.ws_tab, t.ct_tab))
    beta = 0.5 * (1 + math.sqrt(1 - ct)) / math.sqrt(1 - ct)
    sigma = (BC_A_S * ti + BC_B_S) * xd + BC_C_S * math.sqrt(beta)
    n = BC_A_F * math.exp(BC_B_F * xd) + BC_C_F
    C = 2 ** (2 / n - 1) - math.sqrt(2 ** (4 / n - 2) - n * ct / (16 * math.gamma(2 / n) * sigma ** (4 / n)))
    p = blondel_jimenez_power([0.0, xd * D], [0.0, 0.0], ws, 270.0, ti, np.zeros(2)).numpy()
    assert p[0] == pytest.approx(float(np.interp(ws, t.ws_tab, t.power_tab)), rel=1e-12)
    assert p[1] == pytest.approx(float(np.interp(ws * (1 - C), t.ws_tab, t.power_tab)), rel=1e-9)
    assert 0.25 < C < 0.45                                           # a plausible 7 D deficit
    # yawing the front turbine by 25 deg: Jimenez' skew angle integrated over 7 D (closed form of the quadrature's
    # integrand for small angles), deficit evaluated off-centre
    yaw = 25.0
    g = math.radians(yaw)
    ct_y = float(np.interp(ws * math.cos(g), t.ws_tab, t.ct_tab)) * math.cos(g) ** 2
    a0 = math.cos(g) ** 2 * math.sin(g) * ct_y / 2
    defl = a0 * xd * D / (1 + JIMENEZ_BETA * xd)                     # int_0^x a0 / (1 + beta x'/D)^2 dx'
    p_y = blondel_jimenez_power([0.0, xd * D], [0.0, 0.0], ws, 270.0, ti, np.array([yaw, 0.0])).numpy()
    beta_y = 0.5 * (1 + math.sqrt(1 - ct_y)) / math.sqrt(1 - ct_y)
    sig_y = (BC_A_S * ti + BC_B_S) * xd + BC_C_S * math.sqrt(beta_y)
    C_y = 2 ** (2 / n - 1) - math.sqrt(2 ** (4 / n - 2) - n * ct_y / (16 * math.gamma(2 / n) * sig_y ** (4 / n)))
    u1 = ws * (1 - C_y * math.exp(-(defl / D) ** n / (2 * sig_y ** 2)))
    assert p_y[1] == pytest.approx(float(np.interp(u1, t.ws_tab, t.power_tab)), rel=5e-3)    # sin(a) ~ a, quadrature
    assert 0.3 * D < defl < 0.8 * D and p_y[1] > p[1] * 1.2                                   # steering pays downstream


def test_pywake_agent_reference_test():
    """tests/test_pywake_agent.py:11-45 of the reference, on the restated model."""
    from windgym_amd.steady import PyWakeAgent
    agent = PyWakeAgent(x_pos=[0, 500], y_pos=[0, 0], wind_speed=6, wind_dir=270, TI=0.02)
    nominal = agent.power([30, 0])
    agent.optimize()
    assert agent.power(agent.optimized_yaws) >= nominal
    assert agent.model == "blondel_jimenez" and agent.power(agent.optimized_yaws) > agent.power([0, 0])


# ---- the agents' batched torch evaluation against the CPU restatement under oracle/ (the checker k_steady is held to as well) ----
@pytest.mark.parametrize("model", ["m0", "blondel_jimenez"])
def test_torch_evaluation_matches_the_oracle_restatement(model):
    from oracle import steady_oracle as so
    from windgym_amd import steady
    from windgym_amd.config import rotor_points
    from windgym_amd.turbine import as_tabular
    x, y = np.meshgrid(np.linspace(0, 1280, 4), np.linspace(0, 853.3, 3))
    x, y = x.ravel(), y.ravel()
    rng = np.random.default_rng(3)
    C = 9
    ws = rng.uniform(6.0, 14.0, C); wd = rng.uniform(240.0, 300.0, C); ti = rng.uniform(0.03, 0.12, C)
    yaw = rng.uniform(-30.0, 30.0, (C, len(x)))
    tab = as_tabular(V80())
    D = float(tab.diameter())
    ry, rz = rotor_points(16, 0.5 * D)
    fn = steady.steady_state_power if model == "m0" else steady.blondel_jimenez_power
    got = fn(x, y, ws, wd, ti, yaw).numpy()
    for c in range(C):
        if model == "m0":
            ref = so.m0_steady_power(x, y, ws[c], wd[c], ti[c], yaw[c], tab.ws_tab, tab.power_tab, tab.ct_tab, D, ry, rz)
        else:
            ref = so.blondel_jimenez_power(x, y, ws[c], wd[c], ti[c], yaw[c], tab.ws_tab, tab.power_tab, tab.ct_tab, D)
        np.testing.assert_allclose(got[c], ref, rtol=1e-9, atol=1e-3)
