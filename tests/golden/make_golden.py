#!/usr/bin/env python3
"""Generate golden vectors for the WindGym *glue* (everything on the step() path except the
external DYNAMIKS physics) by importing the reference from /root/reference in THIS container.

Run only in the dev container (the reference tree does not exist on the GPU box):

    python tests/golden/make_golden.py

What it does (SURVEY.md Appendix C recipe):
  * stubs `gymnasium`, `pettingzoo`, `dynamiks`, `py_wake`, `IPython` in sys.modules,
  * imports the reference's MesClass / WindEnv / BasicControllers / Wind_Farm_Env / WindEnvMulti
    modules unmodified,
  * replaces the flow simulation by a *scripted double*: every `fs.step()` consumes the next row of
    a pre-drawn (u, v, w, power) table, so that the reference's yaw actuation, baseline controllers,
    measurement averaging, MesClass windows/TI/scaling, rewards, penalties and truncation arithmetic
    are exercised exactly as in `WindFarmEnv.reset()/step()` (Wind_Farm_Env.py:680-1034),
  * records inputs (config, seeds, script tables, actions) and outputs (obs, reward, truncated,
    yaws, sampled wind conditions, time_max, ...) into small .npz fixtures next to this file.

Only data (inputs / expected outputs) is written; no reference source text is stored.
"""
import importlib
import json
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import yaml

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

# ----------------------------------------------------------------------------------------------
# V80 tabular turbine (py_wake hornsrev1 V80: D=80 m, hub 70 m); same table as windgym_amd.turbine
# ----------------------------------------------------------------------------------------------
V80_WS = np.arange(3.0, 26.0, 1.0)
V80_P = 1e3 * np.array([0, 66.6, 154, 282, 460, 696, 996, 1341, 1661, 1866, 1958, 1988, 1997, 1999]
                       + [2000] * 9, dtype=float)
V80_CT = np.array([0, .818, .806, .804, .805, .806, .807, .793, .739, .709, .409, .314, .249, .202,
                   .167, .140, .119, .102, .088, .077, .067, .060, .053])


class FakeTurbine:
    def power(self, ws):
        return np.interp(ws, V80_WS, V80_P, left=0.0, right=0.0)

    def diameter(self):
        return 80.0

    def hub_height(self):
        return 70.0


# ----------------------------------------------------------------------------------------------
# stubs
# ----------------------------------------------------------------------------------------------
def install_stubs():
    gym = types.ModuleType("gymnasium")

    class Env:
        _np_random = None

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random = np.random.default_rng()
            return self._np_random

        def reset(self, seed=None, options=None):
            if seed is not None:
                self._np_random = np.random.default_rng(seed)

        @property
        def unwrapped(self):
            return self

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    spaces = types.ModuleType("gymnasium.spaces")
    spaces.Box = Box
    gym.Env = Env
    gym.spaces = spaces
    sys.modules["gymnasium"] = gym
    sys.modules["gymnasium.spaces"] = spaces
    pz = types.ModuleType("pettingzoo")

    class ParallelEnv:
        # The reference's WindFarmEnvMulti.__init__ calls reset() (via WindFarmEnv.__init__, reset_init)
        # before `possible_agents` exists (WindEnvMulti.py:42-72 vs :151) and raises AttributeError with
        # the real pettingzoo base class.  Class-level defaults let the constructor finish so that the
        # per-agent packing code that follows can be recorded.
        possible_agents = []
        agents = []

    pz.ParallelEnv = ParallelEnv
    sys.modules["pettingzoo"] = pz
    for name in [
        "dynamiks", "dynamiks.dwm", "dynamiks.dwm.particle_deficit_profiles",
        "dynamiks.dwm.particle_deficit_profiles.ainslie", "dynamiks.dwm.particle_motion_models",
        "dynamiks.sites", "dynamiks.sites.turbulence_fields", "dynamiks.wind_turbines",
        "dynamiks.wind_turbines.hawc2_windturbine", "dynamiks.views",
        "dynamiks.dwm.added_turbulence_models", "IPython", "py_wake", "py_wake.wind_turbines",
    ]:
        sys.modules[name] = MagicMock()
    pkg = types.ModuleType("WindGym")
    pkg.__path__ = [os.path.join(REF, "WindGym")]
    sys.modules["WindGym"] = pkg


# ----------------------------------------------------------------------------------------------
# scripted flow-simulation double
# ----------------------------------------------------------------------------------------------
SCRIPTS = {}      # farm_id -> dict(uvw=[T,N,3], power=[T,N])
_FS_COUNT = [0]   # how many DWMFlowSimulation doubles were built since the last reset of the counter


class FakeWTs:
    def __init__(self, x, y, windTurbine=None):
        self.x = np.asarray(x, dtype=float)
        self.y = np.asarray(y, dtype=float)
        self._yaw = np.zeros(len(self.x))
        self.fs = None
        self.types = 0

    # DYNAMIKS exposes `yaw` as a sensor-backed property: assigning copies the values.  (A plain attribute
    # would alias fs.windTurbines.yaw and fs_baseline.windTurbines.yaw after Wind_Farm_Env.py:781 and let
    # the in-place `+=` of :832 leak the agent's action into the baseline farm.)
    @property
    def yaw(self):
        return self._yaw

    @yaw.setter
    def yaw(self, value):
        self._yaw = np.array(value, dtype=float)

    @property
    def rotor_avg_windspeed(self):
        return self.fs.script["uvw"][self.fs.idx].copy()

    def power(self):
        return self.fs.script["power"][self.fs.idx].copy()

    def _rot(self):
        # flow frame: x downwind; rotation by theta = 270 - wd about the farm centre (model M0)
        th = np.deg2rad(270.0 - self.fs.wind_direction)
        cx, cy = self.x.mean(), self.y.mean()
        dx, dy = self.x - cx, self.y - cy
        xr = cx + dx * np.cos(th) + dy * np.sin(th)
        yr = cy - dx * np.sin(th) + dy * np.cos(th)
        return xr, yr

    @property
    def positions_xyz(self):
        xr, yr = self._rot()
        return np.array([xr, yr, np.full_like(xr, 70.0)])

    @property
    def rotor_positions_xyz(self):
        return self.positions_xyz


class FakeFS:
    def __init__(self, site=None, windTurbines=None, wind_direction=270.0, dt=1, **kw):
        self.windTurbines = windTurbines
        windTurbines.fs = self
        self.wind_direction = wind_direction
        self.dt = dt
        self.time = 0
        self.idx = 0
        self.n_steps = 0
        self.n_run = 0
        farm_id = _FS_COUNT[0] % 2 if SCRIPTS.get("two_farms") else 0
        _FS_COUNT[0] += 1
        # each new episode continues reading the script where the previous episode of that farm stopped
        self.farm_id = farm_id
        self.script = SCRIPTS[farm_id]
        self.idx = SCRIPTS["cursor"][farm_id]

    def step(self):
        self.idx += 1
        SCRIPTS["cursor"][self.farm_id] = self.idx
        self.time += self.dt
        self.n_steps += 1

    def run(self, t):
        self.time += t
        self.n_run += t


class FakeSite:
    def __init__(self, ws=None, turbulenceField=None):
        self.ws = ws


def import_reference():
    install_stubs()
    mods = {}
    for m in ["WindEnv", "MesClass", "BasicControllers", "Wind_Farm_Env", "WindEnvMulti"]:
        mods[m] = importlib.import_module("WindGym." + m)
    wfe = mods["Wind_Farm_Env"]
    wfe.DWMFlowSimulation = FakeFS
    wfe.TurbulenceFieldSite = FakeSite
    wfe.PyWakeWindTurbines = FakeWTs
    return mods


# ----------------------------------------------------------------------------------------------
# configs (our own YAML documents in the reference's schema; values of the shipped examples)
# ----------------------------------------------------------------------------------------------
def base_cfg(**over):
    cfg = dict(
        yaw_init="Random", noise="None", BaseController="Local", ActionMethod="wind",
        Track_power=False,
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
        yaw_mes=dict(yaw_current=False, yaw_rolling_mean=True, yaw_history_N=1,
                     yaw_history_length=10, yaw_window_length=10),
        power_mes=dict(power_current=False, power_rolling_mean=False, power_history_N=1,
                       power_history_length=10, power_window_length=10),
    )
    for k, v in over.items():
        if isinstance(v, dict) and k in cfg:
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


def cfg_env1():
    return base_cfg()


def cfg_2turb():
    return base_cfg(
        yaw_init="Zeros", ActionMethod="yaw",
        farm=dict(nx=2, ny=1),
        wind=dict(ws_min=6, ws_max=10, TI_min=0.03, wd_min=260, wd_max=280),
        power_def=dict(Power_avg=1),
        ws_mes=dict(ws_history_N=100, ws_history_length=100, ws_window_length=1),
        wd_mes=dict(wd_history_length=10, wd_window_length=10),
        yaw_mes=dict(yaw_rolling_mean=False, yaw_history_N=100, yaw_history_length=100,
                     yaw_window_length=1),
    )


def cfg_4turb():
    return base_cfg(
        yaw_init="Zeros", ActionMethod="yaw",
        wind=dict(ws_min=6, TI_min=0.03, wd_min=270, wd_max=360),
        power_def=dict(Power_avg=1),
        ws_mes=dict(ws_history_length=10, ws_window_length=10),
        wd_mes=dict(wd_history_length=10, wd_window_length=10),
        yaw_mes=dict(yaw_rolling_mean=False),
    )


def cfg_allon(**over):
    """Every sensor on, several windows, both 'current' and 'rolling' -> exercises all MesClass quirks."""
    cfg = base_cfg(
        yaw_init="Random", ActionMethod="yaw",
        farm=dict(nx=3, ny=2, xDist=5, yDist=3),
        mes_level=dict(turb_ws=True, turb_wd=True, turb_TI=True, turb_power=True,
                       farm_ws=True, farm_wd=True, farm_TI=True, farm_power=True),
        ws_mes=dict(ws_current=True, ws_rolling_mean=True, ws_history_N=4, ws_history_length=30,
                    ws_window_length=5),
        wd_mes=dict(wd_current=True, wd_rolling_mean=True, wd_history_N=3, wd_history_length=12,
                    wd_window_length=4),
        yaw_mes=dict(yaw_current=True, yaw_rolling_mean=True, yaw_history_N=2, yaw_history_length=20,
                     yaw_window_length=1),
        power_mes=dict(power_current=False, power_rolling_mean=True, power_history_N=5,
                       power_history_length=17, power_window_length=3),
        power_def=dict(Power_reward="Power_avg", Power_avg=7, Power_scaling=2.5),
        act_pen=dict(action_penalty=0.3, action_penalty_type="Change"),
    )
    for k, v in over.items():
        if isinstance(v, dict) and k in cfg:
            cfg[k].update(v)
        else:
            cfg[k] = v
    return cfg


# ----------------------------------------------------------------------------------------------
def draw_script(rng, T, N):
    """Plausible but arbitrary rotor wind vectors and powers (the glue does not care about physics)."""
    u = rng.uniform(4.0, 16.0, size=(T, N))
    v = rng.normal(0.0, 0.8, size=(T, N))
    w = rng.normal(0.0, 0.4, size=(T, N))
    p = rng.uniform(0.0, 2.0e6, size=(T, N))
    return dict(uvw=np.stack([u, v, w], axis=-1), power=p)


def run_case(mods, name, cfg, kwargs, n_steps, seed, script_seed, multi=False, n_episodes=1,
             action_kind="uniform"):
    wfe = mods["Wind_Farm_Env"]
    nx, ny = cfg["farm"]["nx"], cfg["farm"]["ny"]
    N = nx * ny
    two = cfg["power_def"]["Power_reward"] == "Baseline" or kwargs.get("Baseline_comp", False)
    T = 4000
    rng = np.random.default_rng(script_seed)
    SCRIPTS.clear()
    SCRIPTS[0] = draw_script(rng, T, N)
    SCRIPTS[1] = draw_script(rng, T, N)
    SCRIPTS["two_farms"] = two
    SCRIPTS["cursor"] = {0: 0, 1: 0}
    _FS_COUNT[0] = 0

    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
        ypath = f.name
    cls = mods["WindEnvMulti"].WindFarmEnvMulti if multi else wfe.WindFarmEnv
    kw = dict(turbine=FakeTurbine(), yaml_path=ypath, turbtype="None", seed=seed)
    kw.update(kwargs)
    if not multi:
        kw["reset_init"] = False
    env = cls(**kw)
    arng = np.random.default_rng(script_seed + 1)

    rec = dict(obs0=[], obs=[], reward=[], truncated=[], yaw=[], yaw_base=[], action=[],
               ws=[], wd=[], ti=[], time_max=[], ep_start=[], cursor0=[], cursor1=[],
               n_run=[], n_dev_steps=[], power_agent=[], obs_multi0=[], obs_multi=[])
    step_total = 0
    for ep in range(n_episodes):
        c0 = dict(SCRIPTS["cursor"])
        out = env.reset(seed=seed) if ep == 0 else env.reset()
        if multi:
            obs_m, _ = out
            rec["obs_multi0"].append(np.stack([obs_m[a] for a in env.possible_agents]))
            rec["obs0"].append(np.asarray(wfe.WindFarmEnv._get_obs(env)))
        else:
            rec["obs0"].append(np.asarray(out[0]))
        rec["ws"].append(env.ws), rec["wd"].append(env.wd), rec["ti"].append(env.ti)
        rec["time_max"].append(env.time_max)
        rec["ep_start"].append(step_total)
        rec["cursor0"].append(c0[0]), rec["cursor1"].append(c0[1])
        rec["n_run"].append(env.fs.n_run)
        rec["n_dev_steps"].append(env.fs.n_steps)
        rec.setdefault("yaw_init", []).append(np.array(env.fs.windTurbines.yaw, dtype=float))
        for i in range(n_steps):
            if action_kind == "uniform":
                a = arng.uniform(-1, 1, size=N).astype(np.float32)
            elif action_kind == "ones":
                a = np.ones(N, dtype=np.float32)
            elif action_kind == "const":
                a = np.linspace(-0.8, 0.9, N).astype(np.float32)
            if multi:
                act = {ag: np.array([a[j]], dtype=np.float32) for j, ag in enumerate(env.possible_agents)}
                agents = list(env.agents)
                try:
                    obs_m, rew_m, term_m, trunc_m, _ = env.step(act)
                except AttributeError:
                    # reference bug: at truncation WindFarmEnv.step deletes farm_measurements
                    # (Wind_Farm_Env.py:1019-1022) before WindFarmEnvMulti.step re-reads them
                    # (WindEnvMulti.py:201) -> the multi-agent env raises instead of truncating.
                    rec["raised_at"] = [step_total]
                    break
                obs = np.stack([obs_m[ag] for ag in agents])
                rec["obs_multi"].append(obs)
                reward = rew_m[agents[0]]
                truncated = trunc_m[agents[0]]
                rec["obs"].append(np.zeros(0, dtype=np.float32))
            else:
                obs, reward, terminated, truncated, info = env.step(a)
                rec["obs"].append(np.asarray(obs))
                rec["power_agent"].append(float(info["Power agent"]))
            rec["action"].append(a)
            rec["reward"].append(float(reward))
            rec["truncated"].append(bool(truncated))
            step_total += 1
            if truncated:
                # fs was torn down by the reference (Wind_Farm_Env.py:1003-1023)
                rec["yaw"].append(np.full(N, np.nan))
                rec["yaw_base"].append(np.full(N, np.nan))
                break
            rec["yaw"].append(np.array(env.fs.windTurbines.yaw, dtype=float))
            rec["yaw_base"].append(np.array(env.fs_baseline.windTurbines.yaw, dtype=float)
                                   if two else np.zeros(N))
    os.unlink(ypath)
    out = {k: np.array(v) for k, v in rec.items() if len(v)}
    out["script0_uvw"] = SCRIPTS[0]["uvw"][: SCRIPTS["cursor"][0] + 2]
    out["script0_power"] = SCRIPTS[0]["power"][: SCRIPTS["cursor"][0] + 2]
    out["script1_uvw"] = SCRIPTS[1]["uvw"][: SCRIPTS["cursor"][1] + 2]
    out["script1_power"] = SCRIPTS[1]["power"][: SCRIPTS["cursor"][1] + 2]
    out["x_pos"] = np.asarray(env.x_pos, dtype=float)
    out["y_pos"] = np.asarray(env.y_pos, dtype=float)
    out["obs_var"] = np.array(env.obs_var)
    out["hist_max"] = np.array(env.hist_max)
    out["steps_on_reset"] = np.array(env.steps_on_reset)
    kwj = {k: v for k, v in kwargs.items()}
    out["meta"] = np.array(json.dumps(dict(name=name, cfg=cfg, kwargs=kwj, seed=seed, multi=multi,
                                            n_steps=n_steps, n_episodes=n_episodes,
                                            two_farms=bool(two))))
    path = os.path.join(OUT, f"glue_{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name:28s} N={N} O={int(env.obs_var)} steps={step_total} two_farms={two} -> {os.path.basename(path)}"
          f" ({os.path.getsize(path) / 1024:.0f} KiB)")


def fuzz_cfg(rng):
    """A random but valid YAML document + constructor kwargs: every switch of the glue gets exercised in
    combinations the hand-written scenarios do not cover."""
    def mes(prefix):
        hlen = int(rng.integers(1, 40))
        win = int(rng.integers(1, hlen + 1))
        cur, rol = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if not cur and not rol:
            rol = True
        return {f"{prefix}_current": cur, f"{prefix}_rolling_mean": rol,
                f"{prefix}_history_N": int(rng.integers(1, 6)), f"{prefix}_history_length": hlen,
                f"{prefix}_window_length": win}
    flags = {k: bool(rng.integers(0, 2)) for k in ("turb_ws", "turb_wd", "turb_TI", "turb_power", "farm_ws",
                                                   "farm_wd", "farm_TI", "farm_power")}
    if not any(flags.values()):
        flags["turb_ws"] = True
    reward = str(rng.choice(["Baseline", "Power_avg", "None", "Power_diff"]))
    pavg = int(rng.integers(40, 70)) if reward == "Power_diff" else int(rng.integers(1, 30))
    ymin = -float(rng.choice([10, 25, 45]))
    cfg = base_cfg(
        yaw_init=str(rng.choice(["Zeros", "Random"])), BaseController=str(rng.choice(["Local", "Global"])),
        ActionMethod=str(rng.choice(["yaw", "wind"])),
        farm=dict(yaw_min=ymin, yaw_max=float(rng.choice([15, 30, 45])), xDist=float(rng.choice([3, 4, 6.5])),
                  yDist=float(rng.choice([3, 4, 5])), nx=int(rng.integers(1, 4)), ny=int(rng.integers(1, 3))),
        wind=dict(ws_min=float(rng.uniform(5, 9)), ws_max=float(rng.uniform(9, 16)), TI_min=0.02, TI_max=0.15,
                  wd_min=float(rng.uniform(240, 268)), wd_max=float(rng.uniform(272, 300))),
        act_pen=dict(action_penalty=float(rng.choice([0.0, 0.0005, 0.05, 0.4])),
                     action_penalty_type=str(rng.choice(["Change", "Total"]))),
        power_def=dict(Power_reward=reward, Power_avg=pavg, Power_scaling=float(rng.choice([1.0, 0.5, 3.0]))),
        mes_level=flags, ws_mes=mes("ws"), wd_mes=mes("wd"), yaw_mes=mes("yaw"), power_mes=mes("power"),
    )
    fill = [True, False, int(rng.integers(2, 9))][int(rng.integers(0, 3))]
    kwargs = dict(n_passthrough=int(rng.integers(1, 3)), fill_window=fill, dt_sim=1, dt_env=int(rng.choice([1, 1, 2])),
                  yaw_step=float(rng.choice([0.5, 1, 2.5])), Baseline_comp=bool(rng.integers(0, 2)),
                  TI_min_mes=float(rng.choice([0.0, 0.01])), TI_max_mes=float(rng.choice([0.5, 0.3])))
    return cfg, kwargs


def fuzz_cases(mods, n=16):
    rng = np.random.default_rng(20260929)
    made = 0
    tries = 0
    while made < n and tries < 200:
        tries += 1
        cfg, kwargs = fuzz_cfg(rng)
        try:
            run_case(mods, f"fuzz{made:02d}", cfg, kwargs, int(rng.integers(30, 90)), seed=int(rng.integers(1, 10**6)),
                     script_seed=500 + tries, n_episodes=int(rng.choice([1, 1, 2])),
                     action_kind=str(rng.choice(["uniform", "uniform", "const"])))
            made += 1
        except (ValueError, NotImplementedError, ZeroDivisionError) as e:      # combination the reference rejects
            print(f"fuzz try {tries}: reference raised {type(e).__name__}: {e}")
    assert made == n


def mes_unit_cases(mods):
    """Known answers of the bare `Mes` window logic for a sweep of (history_N, window, length, count)."""
    Mes = mods["MesClass"].Mes
    rng = np.random.default_rng(7)
    rows = []
    for hist_n in (1, 2, 3, 4, 7):
        for win in (1, 2, 5, 10):
            for hlen in (1, 5, 10, 30):
                for cur in (False, True):
                    m = Mes(current=cur, rolling_mean=True, history_N=hist_n, history_length=hlen,
                            window_length=win)
                    vals = rng.uniform(0, 30, size=hlen + 7)
                    for n, v in enumerate(vals):
                        m.add_measurement(float(v))
                        if n + 1 in (1, 2, 3, win - 1, win, win + 1, hlen - 1, hlen, hlen + 5):
                            got = m.get_measurements()
                            rows.append(dict(hist_n=hist_n, win=win, hlen=hlen, cur=cur,
                                             vals=vals[: n + 1].tolist(), out=got.astype(float).tolist()))
    with open(os.path.join(OUT, "mes_windows.json"), "w") as f:
        json.dump(rows, f)
    print(f"mes_windows.json: {len(rows)} known-answer rows")


def record_episode_vals_case():
    """Golden vector for the reference's RecordEpisodeVals (wrappers/recordEpisodeVals.py:8-64), the host-side form of
    the episode metrics that get all-reduced across GPUs.

    The wrapper subclasses gymnasium.wrappers.vector.RecordEpisodeStatistics, which is not installed here.  The base
    class below restates the bookkeeping of gymnasium 1.x that the subclass relies on (num_envs, prev_dones,
    episode_returns / episode_lengths with the NEXT-step autoreset convention: the step after a done is the reset step
    and is not counted); the subclass itself is the reference's file, imported unmodified.  The vector env is a scripted
    double: rewards, infos["Power agent"] and dones come from pre-drawn tables."""
    from collections import deque
    gym = sys.modules["gymnasium"]

    class RecordEpisodeStatistics:
        def __init__(self, env, buffer_length=100, stats_key="episode"):
            self.env = env
            self.num_envs = env.num_envs
            self.episode_returns = np.zeros(())
            self.episode_lengths = np.zeros((), dtype=int)
            self.prev_dones = np.zeros((), dtype=bool)
            self.return_queue = deque(maxlen=buffer_length)
            self.length_queue = deque(maxlen=buffer_length)

        def reset(self, seed=None, options=None):
            obs, info = self.env.reset(seed=seed, options=options)
            self.episode_returns = np.zeros(self.num_envs)
            self.episode_lengths = np.zeros(self.num_envs, dtype=int)
            self.prev_dones = np.zeros(self.num_envs, dtype=bool)
            return obs, info

        def step(self, actions):
            obs, rewards, terminations, truncations, infos = self.env.step(actions)
            self.episode_returns[self.prev_dones] = 0
            self.episode_returns[~self.prev_dones] += rewards[~self.prev_dones]
            self.episode_lengths[self.prev_dones] = 0
            self.episode_lengths[~self.prev_dones] += 1
            self.prev_dones = dones = np.logical_or(terminations, truncations)
            if np.sum(dones):
                for i in np.where(dones):
                    self.return_queue.extend(self.episode_returns[i])
                    self.length_queue.extend(self.episode_lengths[i])
            return obs, rewards, terminations, truncations, infos

    wrappers = types.ModuleType("gymnasium.wrappers")
    wvec = types.ModuleType("gymnasium.wrappers.vector")
    wvec.RecordEpisodeStatistics = RecordEpisodeStatistics
    wrappers.vector = wvec
    gym.wrappers = wrappers
    vec = types.ModuleType("gymnasium.vector")
    vvenv = types.ModuleType("gymnasium.vector.vector_env")
    vvenv.ArrayType = np.ndarray
    vvenv.VectorEnv = object
    vec.vector_env = vvenv
    core = types.ModuleType("gymnasium.core")
    core.ActType = object
    core.ObsType = object
    gym.core = core
    for name, mod in (("gymnasium.wrappers", wrappers), ("gymnasium.wrappers.vector", wvec), ("gymnasium.vector", vec),
                      ("gymnasium.vector.vector_env", vvenv), ("gymnasium.core", core)):
        sys.modules[name] = mod
    pkg = types.ModuleType("WindGym.wrappers")
    pkg.__path__ = [os.path.join(REF, "WindGym", "wrappers")]
    sys.modules["WindGym.wrappers"] = pkg
    rev = importlib.import_module("WindGym.wrappers.recordEpisodeVals")

    B, T = 6, 400
    rng = np.random.default_rng(2024)
    ep_len = rng.integers(3, 40, size=(B, 64))              # scripted episode lengths per env
    power = rng.uniform(0.2e6, 7.5e6, size=(T, B))
    reward = rng.normal(0.0, 0.3, size=(T, B))
    done = np.zeros((T, B), dtype=bool)                      # next-step autoreset stream: done, then one reset row
    is_reset_row = np.zeros((T, B), dtype=bool)
    for b in range(B):
        t, k = 0, 0
        while True:
            t += int(ep_len[b, k])
            if t - 1 >= T:
                break
            done[t - 1, b] = True
            if t < T:
                is_reset_row[t, b] = True
            t += 1                                           # the reset step
            k += 1

    class ScriptedVecEnv:
        num_envs = B

        def __init__(self):
            self.t = -1

        def reset(self, seed=None, options=None):
            self.t = -1
            return np.zeros((B, 1)), {}

        def step(self, actions):
            self.t += 1
            return (np.zeros((B, 1)), reward[self.t].copy(), np.zeros(B, dtype=bool), done[self.t].copy(),
                    {"Power agent": power[self.t].copy()})

    w = rev.RecordEpisodeVals(ScriptedVecEnv(), buffer_length=10000)
    w.reset()
    q_len = np.zeros(T, dtype=np.int64)
    for t in range(T):
        w.step(None)
        q_len[t] = len(w.mean_power_queue)
    np.savez_compressed(os.path.join(OUT, "record_episode_vals.npz"), power=power, reward=reward, done=done,
                        is_reset_row=is_reset_row, queue_len=q_len,
                        mean_power_queue=np.asarray(w.mean_power_queue, dtype=np.float64),
                        return_queue=np.asarray(w.return_queue, dtype=np.float64),
                        length_queue=np.asarray(w.length_queue, dtype=np.int64))
    print(f"record_episode_vals.npz: {len(w.mean_power_queue)} episodes over {T} steps x {B} envs")


def main():
    mods = import_reference()
    mes_unit_cases(mods)
    record_episode_vals_case()
    # the three shipped example configurations (Baseline reward -> two farms)
    run_case(mods, "env1", cfg_env1(), dict(n_passthrough=2), 400, seed=1, script_seed=100, n_episodes=2)
    run_case(mods, "2turb", cfg_2turb(), dict(n_passthrough=2), 260, seed=3, script_seed=101, n_episodes=2)
    run_case(mods, "4turb", cfg_4turb(), dict(n_passthrough=2), 250, seed=5, script_seed=102)
    # every sensor on; Power_avg reward, "Change" penalty; no fill -> warm-up windows visible
    run_case(mods, "allon_nofill", cfg_allon(), dict(n_passthrough=1, fill_window=False), 60, seed=11,
             script_seed=103)
    run_case(mods, "allon_fill5", cfg_allon(act_pen=dict(action_penalty_type="Total")),
             dict(n_passthrough=1, fill_window=5, Baseline_comp=True), 60, seed=12, script_seed=104)
    # wind action method against the yaw limits (actions pinned at +1) and Global baseline controller
    run_case(mods, "wind_ones_global", base_cfg(BaseController="Global",
                                                farm=dict(yaw_min=-20, yaw_max=25)),
             dict(n_passthrough=1, yaw_step=2.5), 60, seed=21, script_seed=105, action_kind="ones")
    run_case(mods, "yaw_ones", cfg_4turb(), dict(n_passthrough=1, yaw_step=3), 40, seed=22, script_seed=106,
             action_kind="ones")
    # sub-stepping: dt_env = 3 * dt_sim (means over the sub-steps, controller every sub-step)
    run_case(mods, "substeps3", cfg_env1(), dict(n_passthrough=2, dt_sim=1, dt_env=3), 80, seed=31,
             script_seed=107)
    # Power_diff reward (needs Power_avg >= 40) and "None" reward
    run_case(mods, "power_diff", base_cfg(power_def=dict(Power_reward="Power_diff", Power_avg=50),
                                          act_pen=dict(action_penalty=0.05, action_penalty_type="Total")),
             dict(n_passthrough=2), 120, seed=41, script_seed=108)
    run_case(mods, "power_none", base_cfg(power_def=dict(Power_reward="None")),
             dict(n_passthrough=1), 30, seed=42, script_seed=109)
    # PettingZoo facade (per-agent obs packing, double timestep increment)
    run_case(mods, "multi_3x3", base_cfg(farm=dict(nx=3, ny=3),
                                         mes_level=dict(turb_wd=True, farm_ws=True, farm_power=True,
                                                        farm_TI=True),
                                         wd_mes=dict(wd_rolling_mean=True),
                                         power_mes=dict(power_rolling_mean=True),
                                         ws_mes=dict(ws_history_length=60, ws_window_length=25)),
             dict(n_passthrough=2), 300, seed=51, script_seed=110, multi=True)
    run_case(mods, "multi_env1", cfg_env1(), dict(n_passthrough=2), 200, seed=52, script_seed=111,
             multi=True)
    # randomised combinations of every switch of the glue
    fuzz_cases(mods)


if __name__ == "__main__":
    main()
