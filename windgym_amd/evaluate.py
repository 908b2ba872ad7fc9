"""Batched evaluation sweeps — the caller on the other side of the step() path.

The reference evaluates one (ws, wd, TI) condition at a time (`AgentEval.eval_multiple` loops over
`eval_single_fast`, WindGym/AgentEval.py:39-477, 579-617).  Here every condition is one env of a batch, so a
whole sweep is ONE rollout on the GPU.  The result follows the reference's dataset schema
(AgentEval.py:363-475): dims (time, turb, ws, wd, TI, turbbox, model_step), variables powerF_a, powerT_a, yaw_a,
ws_a, reward (+ powerF_b, powerT_b, yaw_b, ws_b, pct_inc with a baseline farm).  An `xarray.Dataset` is returned
when xarray is installed, otherwise a plain dict {"coords": ..., "data": ...}.
"""
from __future__ import annotations

import itertools

import numpy as np

from .envs import WindFarmVecEnv, _np


def eval_sweep(turbine, yaml_path=None, model=None, *, winddirs=(270.0,), windspeeds=(10.0,),
               turbintensities=(0.05,), t_sim=100, turbtype="None", turbbox="Default", model_step=1,
               Baseline_comp=True, yaw_init="Zeros", deterministic=True, seed=1, device=None, flow_script=None,
               turbboxes=None, **env_kwargs):
    """Roll `model` (anything with predict(obs) -> (action, state)) for t_sim steps under every combination of
    the given wind directions, speeds and turbulence intensities at once.  ``flow_script`` = (uvw [F,T,B,N,3],
    power [F,T,B,N]): replay mode (test hook — the golden vectors recorded from the reference's eval_single_fast)."""
    # `turbboxes`: the reference's eval_multiple loops over turbulence-box files too (AgentEval.py:579-617,
    # FarmEval.update_tf).  Every entry — a TF_* file path or a (box [3, Nx, Ny, Nz], spacing) pair — becomes one box of
    # the device-resident pool, and every (condition, box) pair one env pinned to its box (wg_set_box_ids).
    box_names, box_ids = [turbbox], None
    if turbboxes:
        from .mann import load_box
        pool, spacing, box_names = [], None, []
        for k, tb in enumerate(turbboxes):
            bx, sp = load_box(tb) if isinstance(tb, str) else (np.asarray(tb[0], dtype=np.float32), tuple(tb[1]))
            if pool and (bx.shape != pool[0].shape or tuple(sp) != tuple(spacing)):
                raise ValueError("turbboxes: all boxes of a sweep must share one shape and spacing")
            pool.append(bx); spacing = sp
            box_names.append(tb if isinstance(tb, str) else f"box{k}")
        env_kwargs = dict(env_kwargs, turbulence_box=(pool, spacing))
        turbtype = "MannLoad"
    nb = len(box_names)
    conds4 = list(itertools.product(windspeeds, winddirs, turbintensities, range(nb)))
    conds = [c[:3] for c in conds4]
    if turbboxes:
        box_ids = np.array([c[3] for c in conds4], dtype=np.int32)
    B = len(conds)
    env = WindFarmVecEnv(turbine, B, yaml_path=yaml_path, turbtype=turbtype, Baseline_comp=Baseline_comp,
                         yaw_init=yaw_init, never_truncate=True, autoreset=False, seed=seed, device=device,
                         **env_kwargs)
    ws, wd, ti = (np.array(c, dtype=np.float64) for c in zip(*conds))
    env.batch.set_wind(ws=ws, wd=wd, ti=ti)                       # FarmEval.set_wind_vals for the whole batch
    if box_ids is not None:
        env.batch.set_box_ids(box_ids)
    if flow_script is not None:
        env.batch.set_flow_script(*flow_script)
    obs, _ = env.reset(seed=seed)
    if hasattr(model, "UseEnv"):                                   # AgentEval.py:126-129
        model.yaw_max, model.yaw_min, model.env = env.cfg.yaw_max, env.cfg.yaw_min, env
    N, two = env.n_turb, env.cfg.baseline_comp
    b = env.batch
    rec = {k: np.zeros((t_sim, B) + s) for k, s in
           dict(powerF_a=(), powerT_a=(N,), yaw_a=(N,), ws_a=(N,), reward=()).items()}
    if two:
        rec.update({k: np.zeros((t_sim, B) + s) for k, s in dict(powerF_b=(), powerT_b=(N,), yaw_b=(N,), ws_b=(N,)).items()})
    time = np.zeros(t_sim)

    def snapshot(i, reward):
        rec["powerT_a"][i] = _np(b.info("power_turb_agent")); rec["powerF_a"][i] = rec["powerT_a"][i].sum(-1)
        rec["yaw_a"][i] = _np(b.info("yaw_agent"))
        rec["ws_a"][i] = np.linalg.norm(_np(b.info("rotor_uvw_agent")), axis=-1)
        rec["reward"][i] = reward
        time[i] = float(_np(b.info("fs_time"))[0])
        if two:
            rec["powerT_b"][i] = _np(b.info("power_turb_base")); rec["powerF_b"][i] = rec["powerT_b"][i].sum(-1)
            rec["yaw_b"][i] = _np(b.info("yaw_base"))
            rec["ws_b"][i] = np.linalg.norm(_np(b.info("rotor_uvw_base")), axis=-1)

    snapshot(0, 0.0)                                               # AgentEval.py:131-147
    for i in range(1, t_sim):
        action = model.predict(obs, deterministic=deterministic)[0]
        action = np.broadcast_to(np.asarray(action, dtype=np.float32), (B, N))
        obs, reward, _, _, _ = env.step(np.ascontiguousarray(action))
        snapshot(i, np.asarray(reward))
    env.batch.check()
    env.close()

    nws, nwd, nti = len(windspeeds), len(winddirs), len(turbintensities)

    def shape(a):      # [time, B, ...] -> [time, (turb,) ws, wd, TI, turbbox, model_step]
        a = a.reshape((t_sim, nws, nwd, nti, nb) + a.shape[2:])
        if a.ndim == 6:
            a = np.moveaxis(a, 5, 1)
        return a[..., None]

    # variable order of the reference's xr.Dataset (AgentEval.py:375-396 without, :410-455 with a baseline farm)
    order = ["powerF_a", "powerT_a", "yaw_a", "ws_a"] + (["powerF_b", "powerT_b", "yaw_b", "ws_b"] if two else []) + ["reward"]
    data = {k: shape(rec[k]) for k in order}
    if two:
        data["pct_inc"] = (data["powerF_a"] - data["powerF_b"]) / data["powerF_b"] * 100.0    # AgentEval.py:147, 209
    coords = dict(time=time, turb=np.arange(N), ws=np.asarray(windspeeds, float), wd=np.asarray(winddirs, float),
                  TI=np.asarray(turbintensities, float), turbbox=list(box_names), model_step=[model_step])
    dims4 = ("time", "ws", "wd", "TI", "turbbox", "model_step")
    dims5 = ("time", "turb", "ws", "wd", "TI", "turbbox", "model_step")
    try:
        import xarray as xr
        return xr.Dataset({k: (dims5 if v.ndim == 7 else dims4, v) for k, v in data.items()}, coords=coords)
    except Exception:
        return dict(coords=coords, dims={k: (dims5 if v.ndim == 7 else dims4) for k, v in data.items()}, data=data)


class AgentEval:
    """The reference's evaluation driver (AgentEval.py:478-715) on top of :func:`eval_sweep`: ``set_conditions`` collects
    the grid, ``eval_multiple`` runs ALL combinations in one batched rollout (the reference loops over them, :579-617)
    and keeps the result in ``multiple_eval_ds`` with the reference's dims / variable names; ``save_performance`` /
    ``load_performance`` persist it (``.npz`` — xarray / netCDF are not available here).  Plotting is out of scope.

    ``env`` may be a :class:`FarmEval` / :class:`WindFarmEnv` (its turbine, YAML and constructor kwargs are reused) or
    None with ``turbine`` / ``yaml_path`` given explicitly."""

    def __init__(self, env=None, model=None, name="NoName", t_sim=1000, *, turbine=None, yaml_path=None, **env_kwargs):
        self.ws, self.ti, self.wd, self.yaw, self.turbbox = 10.0, 0.05, 270, 0.0, "Default"
        self.t_sim = t_sim
        self.winddirs, self.windspeeds, self.turbintensities, self.turbboxes = [270], [10], [0.05], ["Default"]
        self.multiple_eval, self.multiple_eval_ds = False, None
        self.env, self.model, self.name = env, model, name
        kw = dict(getattr(env, "_kw", {}))
        self._turbine = turbine if turbine is not None else kw.get("turbine")
        self._yaml = yaml_path if yaml_path is not None else kw.get("yaml_path")
        self._env_kwargs = dict(turbtype=kw.get("turbtype", "None"), yaml_dict=kw.get("yaml_dict"),
                                n_passthrough=kw.get("n_passthrough", 5))
        self._env_kwargs.update(env_kwargs)
        if self._turbine is None:
            raise ValueError("AgentEval needs an env (FarmEval / WindFarmEnv) or turbine= and yaml_path=")

    def set_conditions(self, winddirs: list = [], windspeeds: list = [], turbintensities: list = [],
                       turbboxes: list = ["Default"]):
        if winddirs:
            self.winddirs = list(winddirs)
        if windspeeds:
            self.windspeeds = list(windspeeds)
        if turbintensities:
            self.turbintensities = list(turbintensities)
        if turbboxes:
            self.turbboxes = list(turbboxes)

    def eval_multiple(self, save_figs=False, scale_obs=None, debug=False):
        n = len(self.winddirs) * len(self.windspeeds) * len(self.turbintensities) * len(self.turbboxes)
        print("Running for a total of ", n, "simulations.")
        if save_figs or debug:
            raise NotImplementedError("figures / debug plots of AgentEval are not part of this build")
        kw = {k: v for k, v in self._env_kwargs.items() if v is not None}
        is_default = [isinstance(tb, str) and tb == "Default" for tb in self.turbboxes]
        if any(is_default) and not all(is_default):
            raise ValueError("turbboxes mixes the placeholder 'Default' (the env's own inflow) with box files: "
                             "name every box explicitly")
        self.multiple_eval_ds = eval_sweep(self._turbine, self._yaml, self.model, winddirs=self.winddirs,
                                           windspeeds=self.windspeeds, turbintensities=self.turbintensities,
                                           t_sim=self.t_sim, turbbox=self.turbboxes[0],
                                           # (the reference calls set_condition(turbbox=box) for EVERY entry, also a single one: a real path is
                                           # loaded and pinned, only the placeholder "Default" means the env's own inflow)
                                           turbboxes=self.turbboxes if any(not (isinstance(tb, str) and tb == "Default")
                                                                           for tb in self.turbboxes) else None, **kw)
        self.multiple_eval = True
        return self.multiple_eval_ds

    def save_performance(self, path=None):
        if not self.multiple_eval:
            print("It doenst look like you have any data to save my guy")
            return None
        ds = self.multiple_eval_ds
        path = path or (str(self.name) + "_eval.npz")
        if isinstance(ds, dict):
            flat = {f"data__{k}": v for k, v in ds["data"].items()}
            flat.update({f"coord__{k}": np.asarray(v) for k, v in ds["coords"].items()})
            np.savez_compressed(path, **flat)
        else:                                           # an xarray.Dataset when xarray is installed
            ds.to_netcdf(path if path.endswith(".nc") else path + ".nc")
        return path

    def load_performance(self, path):
        z = np.load(path, allow_pickle=False)
        data = {k[6:]: z[k] for k in z.files if k.startswith("data__")}
        coords = {k[7:]: z[k] for k in z.files if k.startswith("coord__")}
        dims4 = ("time", "ws", "wd", "TI", "turbbox", "model_step")
        dims5 = ("time", "turb", "ws", "wd", "TI", "turbbox", "model_step")
        self.multiple_eval_ds = dict(coords=coords, dims={k: (dims5 if v.ndim == 7 else dims4) for k, v in data.items()},
                                     data=data)
        self.multiple_eval = True
        return self.multiple_eval_ds
