"""YAML-schema documents equivalent to the three configurations the reference ships under
examples/EnvConfigs (2turb.yaml, 4turb.yaml, Env1.yaml): same keys, same values, written as Python dicts so
they can be used without a file (``EnvConfig(yaml_dict=...)``) or dumped with ``yaml.safe_dump``."""
from __future__ import annotations

import copy


def env1_config() -> dict:
    """Env1.yaml: 2x2 farm, 'wind' action, Baseline reward, rolling ws(25) + yaw(10) per turbine."""
    return dict(
        yaw_init="Random", noise="None", BaseController="Local", ActionMethod="wind", Track_power=False,
        farm=dict(yaw_min=-45, yaw_max=45, xDist=4, yDist=4, nx=2, ny=2),
        wind=dict(ws_min=7, ws_max=15, TI_min=0.02, TI_max=0.15, wd_min=255, wd_max=285),
        act_pen=dict(action_penalty=0.0, action_penalty_type="Change"),
        power_def=dict(Power_reward="Baseline", Power_avg=10, Power_scaling=1.0),
        mes_level=dict(turb_ws=True, turb_wd=False, turb_TI=False, turb_power=False,
                       farm_ws=False, farm_wd=False, farm_TI=False, farm_power=False),
        ws_mes=dict(ws_current=False, ws_rolling_mean=True, ws_history_N=1, ws_history_length=25,
                    ws_window_length=25),
        wd_mes=dict(wd_current=False, wd_rolling_mean=False, wd_history_N=1, wd_history_length=20,
                    wd_window_length=20),
        yaw_mes=dict(yaw_current=False, yaw_rolling_mean=True, yaw_history_N=1, yaw_history_length=10,
                     yaw_window_length=10),
        power_mes=dict(power_current=False, power_rolling_mean=False, power_history_N=1,
                       power_history_length=10, power_window_length=10),
    )


def _upd(base, **over):
    cfg = copy.deepcopy(base)
    for k, v in over.items():
        if isinstance(v, dict):
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


def two_turb_config() -> dict:
    """2turb.yaml: 2x1 row, 'yaw' action, 100-sample ws history, wd noise."""
    return _upd(env1_config(), yaw_init="Zeros", noise="Normal", ActionMethod="yaw",
                farm=dict(nx=2, ny=1),
                wind=dict(ws_min=6, ws_max=10, TI_min=0.03, wd_min=260, wd_max=280),
                power_def=dict(Power_avg=1),
                ws_mes=dict(ws_history_N=100, ws_history_length=100, ws_window_length=1),
                wd_mes=dict(wd_history_length=10, wd_window_length=10),
                yaw_mes=dict(yaw_rolling_mean=False, yaw_history_N=100, yaw_history_length=100,
                             yaw_window_length=1))


def four_turb_config() -> dict:
    """4turb.yaml: 2x2 farm, 'yaw' action, 10-sample rolling ws."""
    return _upd(env1_config(), yaw_init="Zeros", noise="Normal", ActionMethod="yaw",
                wind=dict(ws_min=6, TI_min=0.03, wd_min=270, wd_max=360),
                power_def=dict(Power_avg=1),
                ws_mes=dict(ws_history_length=10, ws_window_length=10),
                wd_mes=dict(wd_history_length=10, wd_window_length=10),
                yaw_mes=dict(yaw_rolling_mean=False))


def bench_cfg2_config() -> dict:
    """BASELINE.json configs[1] / the headline metric: 4x4 16-turbine grid, yaw-only action, Env1 sensors and
    wind ranges, Baseline reward (two farms per env, as every shipped YAML implies)."""
    return _upd(env1_config(), ActionMethod="yaw", farm=dict(nx=4, ny=4))


def horns_rev1_layout():
    """Horns Rev 1: 80 turbines, 10 columns x 8 rows on a parallelogram, 7 D = 560 m pitch, columns skewed by
    about 7.2 degrees (regenerated analytically; py_wake's wt_x / wt_y table is not available here).  Returns
    (x_pos, y_pos) in metres.  The reference can only express nx x ny rectangles (Wind_Farm_Env.py:246-252);
    the build accepts an explicit layout through the x_pos / y_pos keyword arguments."""
    import numpy as np
    pitch, skew = 560.0, np.deg2rad(7.2)
    col, row = np.meshgrid(np.arange(10), np.arange(8), indexing="xy")
    x = col * pitch + row * pitch * np.sin(skew)
    y = -row * pitch * np.cos(skew)
    return x.ravel().astype(float), (y - y.min()).ravel().astype(float)


def horns_rev_config() -> dict:
    """BASELINE.json configs[2]: Horns Rev 1 (80 turbines), Env1 sensors, yaw action."""
    return _upd(env1_config(), ActionMethod="yaw", farm=dict(nx=10, ny=8))


def multi_3x3_config() -> dict:
    """BASELINE.json configs[3]: 3x3 farm for the per-turbine-agent PettingZoo facade."""
    return _upd(env1_config(), ActionMethod="yaw", farm=dict(nx=3, ny=3))
