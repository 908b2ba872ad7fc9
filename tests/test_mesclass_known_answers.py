"""The known answers asserted by the reference's own MesClass unit tests (tests/test_MesClass.py), replayed through the
oracle (CPU) and through the HIP path (GPU) by feeding scripted sensor values through the whole glue.
Data: tests/golden/mesclass_known_answers.json (inputs / expected outputs only)."""
import json
import math
import os

import numpy as np
import pytest

from helpers import GOLDEN

KA = json.load(open(os.path.join(GOLDEN, "mesclass_known_answers.json")))


def _farm_cfg(n_envs=1):
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    fm = KA["farm_mes"]
    c = fm["channel"]

    def mes(p):
        return {f"{p}_current": c["current"], f"{p}_rolling_mean": c["rolling_mean"], f"{p}_history_N": c["history_N"],
                f"{p}_history_length": c["history_length"], f"{p}_window_length": c["window_length"]}
    d = dict(
        yaw_init="Defined", noise="None", BaseController="Local", ActionMethod="yaw", Track_power=False,
        farm=dict(yaw_min=-45, yaw_max=45, xDist=4, yDist=4, nx=fm["n_turbines"], ny=1),
        wind=dict(ws_min=8, ws_max=8, TI_min=0.05, TI_max=0.05, wd_min=300, wd_max=300),
        act_pen=dict(action_penalty=0.0, action_penalty_type="Change"),
        power_def=dict(Power_reward="None", Power_avg=1, Power_scaling=1.0),
        mes_level=dict(turb_ws=True, turb_wd=True, turb_TI=True, turb_power=True, farm_ws=True, farm_wd=True,
                       farm_TI=True, farm_power=True),
        ws_mes=mes("ws"), wd_mes=mes("wd"), yaw_mes=mes("yaw"), power_mes=mes("power"))
    p0 = fm["pushes"][0]
    return EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None", fill_window=False, n_particles=32, n_rotor_pts=1,
                     never_truncate=True, yaw_step=10, yaw_defined=p0["yaw"], n_envs=n_envs)


def _script(cfg):
    """(u, v, w, power) rows that make the sensors read the pushes of the fixture: ws = |uvw|, wd = wd_global +
    deg(arctan(v / u)) (Wind_Farm_Env.py:485-492)."""
    fm = KA["farm_mes"]
    N, B = cfg.n_turb, cfg.n_envs
    rows = [fm["pushes"][0]] + fm["pushes"]            # reset's fill step consumes row 1 (row 0 = initial state)
    uvw = np.zeros((1, len(rows) + 1, B, N, 3))
    pw = np.zeros((1, len(rows) + 1, B, N))
    for r, push in enumerate(rows + [rows[-1]]):
        for t in range(N):
            ang = math.radians(push["wd"][t] - 300.0)
            uvw[0, r, :, t, 0] = push["ws"][t] * math.cos(ang)
            uvw[0, r, :, t, 1] = push["ws"][t] * math.sin(ang)
            pw[0, r, :, t] = push["power"][t]
    return uvw, pw


def _expected(cfg, scaled):
    """Observation vector assembled from the reference's known answers (+ TI = 0 is not expected here: the two ws
    pushes differ, so TI is taken from the definition std/mean of MesClass.calc_TI)."""
    ex = KA["farm_mes"]["expected_unscaled"]
    ws_hist = np.array([10.0, 12.0])
    ti = float(np.std(ws_hist) / np.mean(ws_hist))
    pmax = cfg.maxturbpower

    def sc(v, lo, hi):
        v = np.asarray(v, dtype=np.float32)
        return (2.0 * (v - np.float32(lo)) / np.float32(hi - lo) - 1.0) if scaled else v
    wd_lo, wd_hi = cfg.wd_min - 5, cfg.wd_max + 5
    turb = np.concatenate([sc(ex["turb_ws"], 2, 25), sc(ex["turb_wd"], wd_lo, wd_hi),
                           sc(ex["turb_yaw"], cfg.yaw_min, cfg.yaw_max), sc([ti], cfg.TI_min_mes, cfg.TI_max_mes),
                           sc(ex["turb_power"], 0, pmax)])
    ti_farm = sc([ti], cfg.TI_min_mes, cfg.TI_max_mes)      # mean of the (scaled) turbine TIs, all equal here
    farm = np.concatenate([sc(ex["farm_ws"], 2, 25), sc(ex["farm_wd"], wd_lo, wd_hi), ti_farm,
                           sc(ex["farm_power"], 0, cfg.n_turb * pmax)])
    v = np.concatenate([turb] * cfg.n_turb + [farm])
    return np.clip(v, -1, 1) if scaled else v


def test_counts_match_the_reference_assertions():
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    cfg = _farm_cfg()
    assert cfg._observed_variables() == KA["farm_mes"]["observed_variables"] and cfg.hist_max == KA["farm_mes"]["max_hist"]
    tm = KA["turb_mes"]

    def mes(p):
        cur, rol, hn, hl, wl = tm[p]
        return {f"{p}_current": cur, f"{p}_rolling_mean": rol, f"{p}_history_N": hn, f"{p}_history_length": hl,
                f"{p}_window_length": wl}
    d = dict(yaw_init="Zeros", noise="None", BaseController="Local", ActionMethod="yaw", Track_power=False,
             farm=dict(yaw_min=-45, yaw_max=45, xDist=4, yDist=4, nx=1, ny=1),
             wind=dict(ws_min=8, ws_max=8, TI_min=0.05, TI_max=0.05, wd_min=270, wd_max=270),
             act_pen=dict(action_penalty=0.0, action_penalty_type="Change"),
             power_def=dict(Power_reward="None", Power_avg=1, Power_scaling=1.0),
             mes_level=dict(turb_ws=True, turb_wd=True, turb_TI=True, turb_power=True, farm_ws=False, farm_wd=False,
                            farm_TI=False, farm_power=False),
             ws_mes=mes("ws"), wd_mes=mes("wd"), yaw_mes=mes("yaw"), power_mes=mes("power"))
    c1 = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None")
    assert c1.turb_observed_variables() == tm["observed_variables"] and c1.hist_max == tm["max_hist"]


def test_oracle_reproduces_the_farm_mes_known_answers(oracle_lib):
    cfg = _farm_cfg()
    o = oracle_lib.Oracle(cfg)
    o.set_flow_script(*_script(cfg))
    o.reset(seeds=[0])
    a = np.ones((1, cfg.n_turb))                      # yaw: -5 -> +5 with yaw_step = 10
    obs, *_ = o.step(a)
    np.testing.assert_allclose(obs[0], _expected(cfg, scaled=True), rtol=0, atol=3e-6)


def test_oracle_reproduces_the_bare_mes_known_answers(oracle_lib):
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    for row in KA["mes"]:
        d = dict(yaw_init="Zeros", noise="None", BaseController="Local", ActionMethod="yaw", Track_power=False,
                 farm=dict(yaw_min=-45, yaw_max=45, xDist=4, yDist=4, nx=1, ny=1),
                 wind=dict(ws_min=8, ws_max=8, TI_min=0.05, TI_max=0.05, wd_min=270, wd_max=270),
                 act_pen=dict(action_penalty=0.0, action_penalty_type="Change"),
                 power_def=dict(Power_reward="None", Power_avg=1, Power_scaling=1.0),
                 mes_level=dict(turb_ws=True, turb_wd=False, turb_TI=False, turb_power=False, farm_ws=False,
                                farm_wd=False, farm_TI=False, farm_power=False),
                 ws_mes=dict(ws_current=row["current"], ws_rolling_mean=row["rolling_mean"], ws_history_N=row["history_N"],
                             ws_history_length=row["history_length"], ws_window_length=row["window_length"]),
                 wd_mes=dict(wd_current=False, wd_rolling_mean=False, wd_history_N=1, wd_history_length=1, wd_window_length=1),
                 yaw_mes=dict(yaw_current=False, yaw_rolling_mean=False, yaw_history_N=1, yaw_history_length=1,
                              yaw_window_length=1),
                 power_mes=dict(power_current=False, power_rolling_mean=False, power_history_N=1,
                                power_history_length=1, power_window_length=1))
        cfg = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None", fill_window=False, n_particles=32, n_rotor_pts=1,
                        never_truncate=True)
        o = oracle_lib.Oracle(cfg)
        vals = np.array(row["vals"], dtype=float)
        T = len(vals) + 2
        uvw = np.zeros((1, T, 1, 1, 3))
        uvw[0, 1:len(vals) + 1, 0, 0, 0] = vals
        uvw[0, 0, 0, 0, 0] = vals[0]
        uvw[0, -1, 0, 0, 0] = vals[-1]
        o.set_flow_script(uvw, np.zeros((1, T, 1, 1)))
        obs = o.reset(seeds=[0])
        for _ in range(len(vals) - 1):
            obs, *_ = o.step(np.zeros((1, 1)))
        expect = np.clip(2.0 * (np.array(row["out"], dtype=np.float32) - np.float32(2.0)) / np.float32(23.0) - 1.0, -1, 1)
        np.testing.assert_allclose(obs[0], expect, rtol=0, atol=2e-6, err_msg=str(row))


@pytest.mark.gpu
def test_hip_reproduces_the_farm_mes_known_answers():
    import torch
    from windgym_amd import binding
    cfg = _farm_cfg(n_envs=3)
    env = binding.HipBatch(cfg)
    env.set_flow_script(*_script(cfg))
    env.reset(seeds=[0, 1, 2])
    obs, *_ = env.step(torch.ones((3, cfg.n_turb), device="cuda"))
    exp_s, exp_u = _expected(cfg, scaled=True), _expected(cfg, scaled=False)
    raw = env.measurements().cpu().numpy()
    for b in range(3):
        np.testing.assert_allclose(obs[b].cpu().numpy(), exp_s, rtol=0, atol=2e-5)
        # the unscaled values are the reference test's numbers themselves
        np.testing.assert_allclose(raw[b], exp_u, rtol=2e-6, atol=2e-4)
    env.check()
