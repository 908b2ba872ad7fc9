"""Pin the oracle's glue against golden vectors recorded from the reference's own code
(tests/golden/make_golden.py): MesClass windows / TI / scaling, yaw actuation, baseline controllers,
rewards, penalties, sampling, truncation and reset arithmetic."""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN, config_from_meta, golden_cases, load_golden, script_tables

CASES = golden_cases()


def _run_case(om, name, precision="f64"):
    g, meta = load_golden(name)
    cfg = config_from_meta(meta)
    o = om.Oracle(cfg, precision)
    uvw, pw = script_tables(g)
    o.set_flow_script(uvw, pw)
    assert o.obs_dim == int(g["obs_var"]) or meta["multi"]
    n_ep = len(g["ep_start"])
    step = 0
    out = dict(obs=[], reward=[], trunc=[], yaw=[], yaw_base=[], obs_multi=[])
    for ep in range(n_ep):
        obs0 = o.reset(seeds=[meta["seed"]] if ep == 0 else None)
        assert o.info("ws_global")[0] == g["ws"][ep]
        assert o.info("wd_global")[0] == g["wd"][ep]
        assert o.info("ti_global")[0] == g["ti"][ep]
        assert int(o.info("time_max")[0]) == int(g["time_max"][ep])
        np.testing.assert_allclose(o.info("yaw_agent")[0], g["yaw_init"][ep], rtol=0, atol=1e-12)
        np.testing.assert_allclose(obs0[0], g["obs0"][ep], rtol=0, atol=2e-6)
        if meta["multi"]:
            np.testing.assert_allclose(o.obs_multi()[0], g["obs_multi0"][ep], rtol=0, atol=2e-6)
        end = g["ep_start"][ep + 1] if ep + 1 < n_ep else len(g["action"])
        while step < end:
            obs, rew, tr, _ = o.step(g["action"][step][None])
            out["obs"].append(obs[0]), out["reward"].append(rew[0]), out["trunc"].append(tr[0])
            out["yaw"].append(o.info("yaw_agent")[0]), out["yaw_base"].append(o.info("yaw_base")[0])
            if meta["multi"]:
                out["obs_multi"].append(o.obs_multi()[0])
            step += 1
    return g, meta, {k: np.array(v) for k, v in out.items()}


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_glue(oracle_lib, name):
    g, meta, out = _run_case(oracle_lib, name)
    n = len(out["reward"])
    assert n == len(g["reward"])
    np.testing.assert_array_equal(out["trunc"], g["truncated"][:n])
    np.testing.assert_allclose(out["reward"], g["reward"][:n], rtol=1e-9, atol=1e-12, equal_nan=True)
    if not meta["multi"]:
        np.testing.assert_allclose(out["obs"], g["obs"][:n], rtol=0, atol=2e-6)
    else:
        np.testing.assert_allclose(out["obs_multi"], g["obs_multi"][:n], rtol=0, atol=2e-6)
    live = ~np.isnan(g["yaw"][:n, 0])
    np.testing.assert_allclose(out["yaw"][live], g["yaw"][:n][live], rtol=0, atol=1e-10)
    if meta["two_farms"]:
        np.testing.assert_allclose(out["yaw_base"][live], g["yaw_base"][:n][live], rtol=0, atol=1e-9)


def test_mes_window_known_answers(oracle_lib):
    """Bare `Mes` semantics (MesClass.py:70-125) through a one-turbine env whose only sensor is ws."""
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    rows = json.load(open(os.path.join(GOLDEN, "mes_windows.json")))
    checked = 0
    for row in rows[::7]:
        cfg_d = dict(
            yaw_init="Zeros", noise="None", BaseController="Local", ActionMethod="yaw", Track_power=False,
            farm=dict(yaw_min=-45, yaw_max=45, xDist=4, yDist=4, nx=1, ny=1),
            wind=dict(ws_min=8, ws_max=8, TI_min=0.05, TI_max=0.05, wd_min=270, wd_max=270),
            act_pen=dict(action_penalty=0.0, action_penalty_type="Change"),
            power_def=dict(Power_reward="None", Power_avg=1, Power_scaling=1.0),
            mes_level=dict(turb_ws=True, turb_wd=False, turb_TI=False, turb_power=False, farm_ws=False,
                           farm_wd=False, farm_TI=False, farm_power=False),
            ws_mes=dict(ws_current=row["cur"], ws_rolling_mean=True, ws_history_N=row["hist_n"],
                        ws_history_length=row["hlen"], ws_window_length=row["win"]),
            wd_mes=dict(wd_current=False, wd_rolling_mean=False, wd_history_N=1, wd_history_length=1,
                        wd_window_length=1),
            yaw_mes=dict(yaw_current=False, yaw_rolling_mean=False, yaw_history_N=1, yaw_history_length=1,
                         yaw_window_length=1),
            power_mes=dict(power_current=False, power_rolling_mean=False, power_history_N=1,
                           power_history_length=1, power_window_length=1))
        cfg = EnvConfig(turbine=V80(), yaml_dict=cfg_d, turbtype="None", fill_window=False, n_particles=32,
                        n_rotor_pts=1, never_truncate=True)
        o = oracle_lib.Oracle(cfg)
        vals = np.array(row["vals"])
        T = len(vals) + 2
        uvw = np.zeros((1, T, 1, 1, 3))
        uvw[0, 1:len(vals) + 1, 0, 0, 0] = vals      # reset's single fill step consumes row 1
        uvw[0, 0, 0, 0, 0] = vals[0]
        uvw[0, -1, 0, 0, 0] = vals[-1]
        o.set_flow_script(uvw, np.zeros((1, T, 1, 1)))
        obs = o.reset(seeds=[0])
        for _ in range(len(vals) - 1):
            obs, *_ = o.step(np.zeros((1, 1)))
        expect = 2.0 * (np.array(row["out"], dtype=np.float32) - np.float32(2.0)) / np.float32(23.0) - 1.0
        expect = np.clip(expect, -1, 1)
        np.testing.assert_allclose(obs[0], expect, rtol=0, atol=2e-6)
        checked += 1
    assert checked > 100
