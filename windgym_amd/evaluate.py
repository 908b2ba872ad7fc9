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
               Baseline_comp=True, yaw_init="Zeros", deterministic=True, seed=1, device=None, **env_kwargs):
    """Roll `model` (anything with predict(obs) -> (action, state)) for t_sim steps under every combination of
    the given wind directions, speeds and turbulence intensities at once."""
    conds = list(itertools.product(windspeeds, winddirs, turbintensities))
    B = len(conds)
    env = WindFarmVecEnv(turbine, B, yaml_path=yaml_path, turbtype=turbtype, Baseline_comp=Baseline_comp,
                         yaw_init=yaw_init, never_truncate=True, autoreset=False, seed=seed, device=device,
                         **env_kwargs)
    ws, wd, ti = (np.array(c, dtype=np.float64) for c in zip(*conds))
    env.batch.set_wind(ws=ws, wd=wd, ti=ti)                       # FarmEval.set_wind_vals for the whole batch
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
        a = a.reshape((t_sim, nws, nwd, nti) + a.shape[2:])
        if a.ndim == 5:
            a = np.moveaxis(a, 4, 1)
        return a[..., None, None]

    data = {k: shape(v) for k, v in rec.items()}
    if two:
        data["pct_inc"] = (data["powerF_a"] - data["powerF_b"]) / data["powerF_b"] * 100.0    # AgentEval.py:147, 209
    coords = dict(time=time, turb=np.arange(N), ws=np.asarray(windspeeds, float), wd=np.asarray(winddirs, float),
                  TI=np.asarray(turbintensities, float), turbbox=[turbbox], model_step=[model_step])
    dims4 = ("time", "ws", "wd", "TI", "turbbox", "model_step")
    dims5 = ("time", "turb", "ws", "wd", "TI", "turbbox", "model_step")
    try:
        import xarray as xr
        return xr.Dataset({k: (dims5 if v.ndim == 7 else dims4, v) for k, v in data.items()}, coords=coords)
    except Exception:
        return dict(coords=coords, dims={k: (dims5 if v.ndim == 7 else dims4) for k, v in data.items()}, data=data)
