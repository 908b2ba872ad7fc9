"""Configuration surface of the reference, flattened for the C ABI.

``EnvConfig`` accepts exactly what ``WindFarmEnv.__init__`` accepts (Wind_Farm_Env.py:50-70) plus the
YAML document read by ``load_config`` (:349-399) and derives everything the constructor derives:
layout (:244-252), ``Baseline_comp`` (:217-220), ``hist_max`` / ``steps_on_reset`` (:225-240), scaling
ranges of the sensors (:409-451) and the observation length.  ``to_c()`` produces the ``wg_config``
struct of include/windgym_hip.h.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import numpy as np
import yaml

from .turbine import as_tabular

WG_ABI_VERSION = 4
WG_N_CH = 4
WG_N_METRICS = 8
CH_NAMES = ("ws", "wd", "yaw", "power")

ACT = {"yaw": 0, "wind": 1}
CTRL = {"Local": 0, "Global": 1}
YAWINIT = {"Zeros": 0, "Random": 1, "Defined": 2}
REW = {"Baseline": 0, "Power_avg": 1, "None": 2, "Power_diff": 3}
PEN = {"Change": 0, "Total": 1}
NOISE = {"None": 0, "Normal": 1}
# turbtype -> inflow mode of the build.  The three Mann variants share one frozen box per GPU
# (DESIGN.md §2.5); "Random" = i.i.d. gusts; "None" = uniform inflow.
TURB = {"None": 0, "Random": 1, "MannFixed": 2, "MannGenerate": 3, "MannLoad": 4}

# wg_info_field
INFO = dict(
    yaw_agent=0, yaw_base=1, ws_global=2, wd_global=3, ti_global=4, ws_turb=5, wd_turb=6,
    power_turb_agent=7, power_turb_base=8, power_agent=9, power_base=10, ws_turb_base=11,
    turb_x=12, turb_y=13, timestep=14, time_max=15, fs_time=16, episode=17, rotor_uvw_agent=18,
    rotor_uvw_base=19, rated_power=20, wind_f64=21, step_power_agent=22, step_power_base=23, box_id=24,
)
INFO_INT = {"timestep", "time_max", "episode", "box_id"}


class CChannel(C.Structure):
    _fields_ = [("current", C.c_int32), ("rolling_mean", C.c_int32), ("history_n", C.c_int32),
                ("history_len", C.c_int32), ("window_len", C.c_int32)]


_DP = C.POINTER(C.c_double)


class CConfig(C.Structure):
    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_envs", C.c_int32), ("n_turb", C.c_int32), ("n_farms", C.c_int32), ("k_sub", C.c_int32),
        ("n_particles", C.c_int32), ("n_rotor_pts", C.c_int32),
        ("dt_sim", C.c_double), ("rotor_diameter", C.c_double), ("hub_height", C.c_double),
        ("d_particle", C.c_double),
        ("x_pos", _DP), ("y_pos", _DP), ("rotor_dy", _DP), ("rotor_dz", _DP),
        ("n_tab", C.c_int32), ("tab_ws", _DP), ("tab_power", _DP), ("tab_ct", _DP),
        ("yaw_min", C.c_double), ("yaw_max", C.c_double), ("yaw_step", C.c_double), ("yaw_start", C.c_double),
        ("action_method", C.c_int32), ("base_controller", C.c_int32), ("yaw_init", C.c_int32),
        ("yaw_defined", _DP),
        ("ws_min", C.c_double), ("ws_max", C.c_double), ("ti_min", C.c_double), ("ti_max", C.c_double),
        ("wd_min", C.c_double), ("wd_max", C.c_double),
        ("n_passthrough", C.c_double), ("never_truncate", C.c_int32),
        ("ch", CChannel * WG_N_CH),
        ("turb_ws", C.c_int32), ("turb_wd", C.c_int32), ("turb_ti", C.c_int32), ("turb_power", C.c_int32),
        ("farm_ws", C.c_int32), ("farm_wd", C.c_int32), ("farm_ti", C.c_int32), ("farm_power", C.c_int32),
        ("ws_scale_min", C.c_double), ("ws_scale_max", C.c_double),
        ("wd_scale_min", C.c_double), ("wd_scale_max", C.c_double),
        ("ti_scale_min", C.c_double), ("ti_scale_max", C.c_double),
        ("power_max", C.c_double),
        ("noise", C.c_int32), ("noise_sigma", C.c_double * WG_N_CH),
        ("reward_mode", C.c_int32), ("power_avg", C.c_int32), ("power_scaling", C.c_double),
        ("action_penalty", C.c_double), ("penalty_type", C.c_int32),
        ("fill_steps_agent", C.c_int32), ("fill_steps_base", C.c_int32), ("autoreset", C.c_int32),
        ("extra_timestep_inc", C.c_int32),
        ("turb_mode", C.c_int32),
        ("m0_ka", C.c_double), ("m0_kb", C.c_double), ("m0_eps", C.c_double), ("m0_hill", C.c_double),
        ("m0_ti_a", C.c_double), ("m0_ti_b", C.c_double), ("m0_ti_c", C.c_double), ("m0_ti_d", C.c_double),
        ("m0_fc_scale", C.c_double),
        ("full_chains", C.c_int32),
        ("added_turbulence", C.c_int32), ("no_ti_fold", C.c_int32), ("deficit_model", C.c_int32),
        ("reserved0_", C.c_int32),
        ("m0_km1", C.c_double), ("m0_km2", C.c_double),
        ("m0_sg_af", C.c_double), ("m0_sg_bf", C.c_double), ("m0_sg_cf", C.c_double),
    ]


def rotor_points(n_pts: int, radius: float):
    """Equal-weight polar quadrature of the rotor disc: rings at the area midpoints of equal-area annuli.

    1 -> hub; 4 -> one ring; 7 -> hub + 6; otherwise rings of 8 (n_pts must then be a multiple of 8,
    e.g. 16 = 2 rings x 8 with staggered azimuths).
    """
    if n_pts == 1:
        return np.zeros(1), np.zeros(1)
    if n_pts == 4:
        rings, per = 1, 4
    elif n_pts == 7:
        ang = np.arange(6) * (2 * np.pi / 6)
        r = radius * math.sqrt(4.0 / 7.0 * 0.5 + 3.0 / 14.0)  # area midpoint of the outer 6/7 annulus
        return np.concatenate([[0.0], r * np.cos(ang)]), np.concatenate([[0.0], r * np.sin(ang)])
    elif n_pts % 8 == 0:
        rings, per = n_pts // 8, 8
    else:
        raise ValueError("n_rotor_pts must be 1, 4, 7 or a multiple of 8")
    dy, dz = [], []
    for i in range(rings):
        r = radius * math.sqrt((i + 0.5) / rings)
        ang = (np.arange(per) + 0.5 * (i % 2)) * (2 * np.pi / per)
        dy.append(r * np.cos(ang))
        dz.append(r * np.sin(ang))
    return np.concatenate(dy), np.concatenate(dz)


def default_particles(x_pos, y_pos, d_particle_m: float) -> int:
    """Ring slots per turbine: the chain must span the farm diagonal for every wind direction."""
    ext = math.hypot(float(np.ptp(x_pos)), float(np.ptp(y_pos)))
    p = int(math.ceil(ext / d_particle_m)) + 8
    return max(32, ((p + 31) // 32) * 32)


@dataclass
class EnvConfig:
    # ---- ctor kwargs of WindFarmEnv (Wind_Farm_Env.py:50-70) ---------------------------------
    turbine: object = None
    n_passthrough: float = 5
    TI_min_mes: float = 0.0
    TI_max_mes: float = 0.50
    TurbBox: str = "Default"
    turbtype: str = "MannLoad"
    yaml_path: Optional[str] = None
    Baseline_comp: bool = False
    yaw_init: Optional[str] = None
    seed: Optional[int] = None
    dt_sim: float = 1
    dt_env: float = 1
    yaw_step: float = 1
    fill_window: object = True
    # ---- build-specific ------------------------------------------------------------------------
    n_envs: int = 1
    n_particles: Optional[int] = None
    n_rotor_pts: int = 16
    autoreset: bool = False
    x_pos: Optional[Sequence[float]] = None      # layout override (reference only does nx x ny grids)
    y_pos: Optional[Sequence[float]] = None
    yaml_dict: Optional[dict] = None             # alternative to yaml_path
    never_truncate: bool = False
    extra_timestep_inc: bool = False
    advect_full_chains: bool = False             # True: no chain pruning (exact flow-field view behind the last row)
    mann_pool: int = 1                           # turbtype "MannGenerate": realisations generated on the device; an episode's
                                                 # seed (Wind_Farm_Env.py:623) picks one of them and an offset into it
    yaw_defined: Optional[Sequence[float]] = None
    # constants of flow model M0, keys of wg_config without the m0_ prefix: ka, kb, eps, hill, ti_a ...
    # ti_d, fc_scale; missing keys keep the documented defaults (a calibrated M0 is configuration, not code)
    model_constants: Optional[dict] = None
    # wake-added small-scale turbulence (reference: addedTurbulenceModel, Wind_Farm_Env.py:618-664): "auto" = the
    # reference's choice (an isotropic Mann field scaled inside the wakes for every turbulent inflow, none for
    # turbtype "None"), "iso" / "none" force it
    added_turbulence: str = "auto"
    wake_ti_fold: bool = True                    # Crespo-Hernandez added TI folded into the emitted particles' k (M0 §2.4)
    deficit: str = "gaussian"                    # "gaussian" (north_star) | "super_gaussian" (Blondel & Cathelain 2020) |
    #                                              "ainslie" (tabulated DWM eddy-viscosity deficit, windgym_amd/ainslie.py)
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        self.load_config(self.yaml_path)
        self.tab = as_tabular(self.turbine)
        # dt handling (:102-107)
        self.sim_steps_per_env_step = int(self.dt_env / self.dt_sim)
        if self.dt_env % self.dt_sim != 0:
            raise ValueError("dt_env must be a multiple of dt_sim")
        self.yaw_start = 15.0                                               # :110
        self.maxturbpower = float(max(self.tab.power(np.arange(10, 25, 1))))  # :112
        self.d_particle = 0.2                                               # :116
        if self.Track_power:
            raise NotImplementedError("The Track_power is not implemented yet")  # :165-168
        if self.power_reward not in REW:
            raise ValueError("The Power_reward must be either Baseline, Power_avg, None or Power_diff")
        if self.power_reward == "Power_diff" and self.power_avg < 40:       # :186-190
            raise ValueError("The Power_avg must be larger then 40 for the Power_diff reward. "
                             "Also it should probably be way larger my guy")
        if self.ActionMethod == "absolute":
            raise NotImplementedError("The absolute method is not implemented yet")  # :860-861
        if self.ActionMethod not in ACT:
            raise ValueError("The ActionMethod must be yaw, wind or absolute")       # :864
        if self.turbtype not in TURB:
            raise ValueError("Invalid turbulence type specified")                    # :668
        self.baseline_comp = bool(self.power_reward == "Baseline" or self.Baseline_comp)  # :217-220
        if self.baseline_comp and self.BaseController not in CTRL:
            raise ValueError("The BaseController must be either Local or Global... For now")  # :314
        # yaw init strategy: ctor kwarg wins over YAML (:148-162)
        yi = self.yaw_init if self.yaw_init is not None else self.yaml_yaw_init
        self.yaw_init_mode = yi if yi in ("Random", "Defined") else "Zeros"
        # layout (:244-252), including the linspace quirk
        self.D = float(self.tab.diameter())
        if self.x_pos is None:
            x = np.linspace(0, self.D * self.xDist * self.nx, self.nx)
            y = np.linspace(0, self.D * self.yDist * self.ny, self.ny)
            xv, yv = np.meshgrid(x, y, indexing="xy")
            self.x_pos, self.y_pos = xv.flatten(), yv.flatten()
        else:
            self.x_pos = np.asarray(self.x_pos, dtype=float).ravel()
            self.y_pos = np.asarray(self.y_pos, dtype=float).ravel()
        self.n_turb = int(len(self.x_pos))
        # sensors
        self.channels = []
        for name, sec in zip(CH_NAMES, (self.ws_mes, self.wd_mes, self.yaw_mes, self.power_mes)):
            self.channels.append(dict(
                current=bool(sec[f"{name}_current"]), rolling_mean=bool(sec[f"{name}_rolling_mean"]),
                history_n=int(sec[f"{name}_history_N"]), history_len=int(sec[f"{name}_history_length"]),
                window_len=int(sec[f"{name}_window_length"])))
        # turb_mes.max_hist ignores the power history (MesClass.py:239-246)
        self.hist_max = max(c["history_len"] for c in self.channels[:3])
        fw = self.fill_window
        if fw is True:                                                       # :229-240
            self.steps_on_reset = self.hist_max
        elif fw is False:
            self.steps_on_reset = 1
        elif isinstance(fw, int) and fw >= 1:
            self.steps_on_reset = min(fw, self.hist_max)
        else:
            raise ValueError("fill_window must be True or a non-negative integer")
        if self.n_particles is None:
            self.n_particles = default_particles(self.x_pos, self.y_pos, self.d_particle * self.D)
        self.obs_var = self._observed_variables()

    # -- load_config (Wind_Farm_Env.py:349-399) ----------------------------------------------------
    def load_config(self, config_path):
        if self.yaml_dict is not None:
            config = self.yaml_dict
        else:
            with open(config_path, "r") as fh:
                config = yaml.safe_load(fh)
        self.yaml_yaw_init = config.get("yaw_init")
        self.noise = config.get("noise")
        self.BaseController = config.get("BaseController")
        self.ActionMethod = config.get("ActionMethod")
        self.Track_power = config.get("Track_power")
        farm = config.get("farm")
        self.yaw_min, self.yaw_max = farm["yaw_min"], farm["yaw_max"]
        self.xDist, self.yDist, self.nx, self.ny = farm["xDist"], farm["yDist"], farm["nx"], farm["ny"]
        wind = config.get("wind")
        self.ws_min, self.ws_max = wind["ws_min"], wind["ws_max"]
        self.TI_min, self.TI_max = wind["TI_min"], wind["TI_max"]
        self.wd_min, self.wd_max = wind["wd_min"], wind["wd_max"]
        self.wd_min_mes, self.wd_max_mes = wind["wd_min"], wind["wd_max"]
        self.act_pen = config.get("act_pen")
        self.power_def = config.get("power_def")
        self.mes_level = config.get("mes_level")
        self.ws_mes, self.wd_mes = config.get("ws_mes"), config.get("wd_mes")
        self.yaw_mes, self.power_mes = config.get("yaw_mes"), config.get("power_mes")
        self.action_penalty = self.act_pen["action_penalty"]
        self.action_penalty_type = self.act_pen["action_penalty_type"]
        self.Power_scaling = self.power_def["Power_scaling"]
        self.power_avg = self.power_def["Power_avg"]
        self.power_reward = self.power_def["Power_reward"]

    def _count(self, ch, level_on):
        c = self.channels[ch]
        return (c["current"] and level_on) + (c["rolling_mean"] and level_on) * c["history_n"]

    def turb_observed_variables(self):
        m = self.mes_level
        return int(self._count(0, m["turb_ws"]) + self._count(1, m["turb_wd"]) + self._count(2, True)
                   + bool(m["turb_TI"]) + self._count(3, m["turb_power"]))

    def farm_observed_variables(self):
        m = self.mes_level
        return int(self._count(0, m["farm_ws"]) + self._count(1, m["farm_wd"]) + bool(m["farm_TI"])
                   + self._count(3, m["farm_power"]))

    def _observed_variables(self):
        return self.turb_observed_variables() * self.n_turb + self.farm_observed_variables()

    def obs_layout(self):
        """Slices of the observation vector: {"turb": {channel: (offset in the turbine block, count)},
        "turb_block": length, "farm": {channel: (absolute offset, count)}} (MesClass.py:328-351, 679-703)."""
        m = self.mes_level
        off, turb = 0, {}
        for ch, name, on in ((0, "ws", m["turb_ws"]), (1, "wd", m["turb_wd"]), (2, "yaw", True)):
            n = int(self._count(ch, on)); turb[name] = (off, n); off += n
        n = int(bool(m["turb_TI"])); turb["TI"] = (off, n); off += n
        n = int(self._count(3, m["turb_power"])); turb["power"] = (off, n); off += n
        base, farm = off * self.n_turb, {}
        for name, n in (("ws", self._count(0, m["farm_ws"])), ("wd", self._count(1, m["farm_wd"])),
                        ("TI", bool(m["farm_TI"])), ("power", self._count(3, m["farm_power"]))):
            farm[name] = (base, int(n)); base += int(n)
        return dict(turb=turb, turb_block=off, farm=farm)

    def multi_declared_obs_var(self):
        """WindFarmEnvMulti.obs_var as *declared* (WindEnvMulti.py:65-69): counts farm yaw entries that
        are never produced."""
        m = self.mes_level
        farm_decl = int(self._count(0, m["farm_ws"]) + self._count(1, m["farm_wd"]) + self._count(2, True)
                        + bool(m["farm_TI"]) + self._count(3, m["farm_power"]))
        return self.turb_observed_variables() + farm_decl

    # -- flatten ------------------------------------------------------------------------------------
    def to_c(self) -> CConfig:
        c = CConfig()
        keep = self._keep

        def arr(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            keep.append(a)
            return a.ctypes.data_as(_DP)

        c.abi_version = WG_ABI_VERSION
        c.n_envs, c.n_turb = int(self.n_envs), self.n_turb
        c.n_farms = 2 if self.baseline_comp else 1
        c.k_sub = self.sim_steps_per_env_step
        c.n_particles, c.n_rotor_pts = int(self.n_particles), int(self.n_rotor_pts)
        c.dt_sim, c.rotor_diameter, c.hub_height = float(self.dt_sim), self.D, float(self.tab.hub_height())
        c.d_particle = self.d_particle
        c.x_pos, c.y_pos = arr(self.x_pos), arr(self.y_pos)
        dy, dz = rotor_points(self.n_rotor_pts, 0.5 * self.D)
        c.rotor_dy, c.rotor_dz = arr(dy), arr(dz)
        c.n_tab = len(self.tab.ws_tab)
        c.tab_ws, c.tab_power, c.tab_ct = arr(self.tab.ws_tab), arr(self.tab.power_tab), arr(self.tab.ct_tab)
        c.yaw_min, c.yaw_max = float(self.yaw_min), float(self.yaw_max)
        c.yaw_step, c.yaw_start = float(self.yaw_step), self.yaw_start
        c.action_method = ACT[self.ActionMethod]
        c.base_controller = CTRL.get(self.BaseController, 0)
        c.yaw_init = YAWINIT[self.yaw_init_mode]
        if self.yaw_defined is not None:
            yd = np.asarray(self.yaw_defined, dtype=float).ravel()
            if yd.size == 1:                       # WindEnv._defined_yaw (WindEnv.py:44-56)
                yd = np.ones(self.n_turb) * yd[0]
            elif yd.size != self.n_turb:
                raise ValueError("So I am pretty sure something has gone wrong here. The specified yaw "
                                 "values are not the right length.")
            c.yaw_defined = arr(yd)
        c.ws_min, c.ws_max = float(self.ws_min), float(self.ws_max)
        c.ti_min, c.ti_max = float(self.TI_min), float(self.TI_max)
        c.wd_min, c.wd_max = float(self.wd_min), float(self.wd_max)
        c.n_passthrough = float(self.n_passthrough)
        c.never_truncate = int(self.never_truncate)
        for i, ch in enumerate(self.channels):
            c.ch[i].current, c.ch[i].rolling_mean = int(ch["current"]), int(ch["rolling_mean"])
            c.ch[i].history_n, c.ch[i].history_len = ch["history_n"], ch["history_len"]
            c.ch[i].window_len = ch["window_len"]
        m = self.mes_level
        c.turb_ws, c.turb_wd = int(bool(m["turb_ws"])), int(bool(m["turb_wd"]))
        c.turb_ti, c.turb_power = int(bool(m["turb_TI"])), int(bool(m["turb_power"]))
        c.farm_ws, c.farm_wd = int(bool(m["farm_ws"])), int(bool(m["farm_wd"]))
        c.farm_ti, c.farm_power = int(bool(m["farm_TI"])), int(bool(m["farm_power"]))
        c.ws_scale_min, c.ws_scale_max = 2.0, 25.0                               # :440-441
        c.wd_scale_min, c.wd_scale_max = self.wd_min_mes - 5, self.wd_max_mes + 5  # :443-444
        c.ti_scale_min, c.ti_scale_max = float(self.TI_min_mes), float(self.TI_max_mes)
        c.power_max = self.maxturbpower
        c.noise = NOISE.get(self.noise, 0)
        for i, s in enumerate((0.0, 2.0, 0.0, 0.0)):                             # MesClass.py:436-439
            c.noise_sigma[i] = s
        c.reward_mode = REW[self.power_reward]
        c.power_avg = int(self.power_avg)
        c.power_scaling = float(self.Power_scaling)
        c.action_penalty = float(self.action_penalty)
        c.penalty_type = PEN.get(self.action_penalty_type, 0)
        c.fill_steps_agent, c.fill_steps_base = int(self.steps_on_reset), int(self.hist_max)
        c.autoreset = int(bool(self.autoreset))
        c.extra_timestep_inc = int(bool(self.extra_timestep_inc))
        c.turb_mode = TURB[self.turbtype]
        c.full_chains = int(bool(self.advect_full_chains))
        if self.added_turbulence not in ("auto", "iso", "none"):
            raise ValueError("added_turbulence must be 'auto', 'iso' or 'none'")
        c.added_turbulence = int(self.turbtype != "None" and self.added_turbulence in ("auto", "iso"))
        c.no_ti_fold = int(not self.wake_ti_fold)
        if self.deficit not in ("gaussian", "super_gaussian", "ainslie"):
            raise ValueError("deficit must be 'gaussian', 'super_gaussian' or 'ainslie'")
        c.deficit_model = ("gaussian", "super_gaussian", "ainslie").index(self.deficit)
        if c.deficit_model == 1:
            # Blondel & Cathelain (2020), Table 2 (py_wake BlondelSuperGaussianDeficit2020 — the model of the reference's
            # PyWakeAgent): characteristic width sigma/D = (0.17 TI + 0.005) x/D + 0.2 sqrt(beta); model_constants override
            c.m0_ka, c.m0_kb, c.m0_eps = 0.17, 0.005, 0.2
        for k, v in (self.model_constants or {}).items():
            if not hasattr(c, "m0_" + k):
                raise ValueError(f"unknown model constant {k!r}")
            setattr(c, "m0_" + k, float(v))
        return c
