#!/usr/bin/env python3
"""Golden vectors for row f3 (the evaluation surface): drives the REFERENCE's own ``eval_single_fast``
(WindGym/AgentEval.py:39-477) and ``FarmEval`` (WindGym/FarmEval.py) — imported from /root/reference in the build
container, never copied — with the scripted flow double of make_golden.py and a scripted model, and records the
dataset it assembles: variables, dims order, coords, ``pct_inc``, the initial snapshot at time[0].  xarray is not
installed, so ``xr.Dataset`` is a recording stand-in that keeps exactly what the reference passes to it.

Run (build container only):  python tests/golden/make_eval_golden.py   -> tests/golden/eval_single_*.npz
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

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg      # noqa: E402


class RecordingDataset:
    """What the reference hands to xr.Dataset(data_vars=..., coords=...)."""

    def __init__(self, data_vars=None, coords=None):
        self.data_vars, self.coords = data_vars, coords


def import_eval():
    mods = mg.import_reference()
    xr = types.ModuleType("xarray")
    xr.Dataset = RecordingDataset
    sys.modules["xarray"] = xr
    for name in ["matplotlib", "matplotlib.pyplot", "dynamiks.visualizers", "dynamiks.visualizers.flow_visualizers",
                 "py_wake.utils", "py_wake.utils.plotting"]:
        sys.modules[name] = MagicMock()
    sys.modules["gymnasium"].Env.close = lambda self: None
    mods["FarmEval"] = importlib.import_module("WindGym.FarmEval")
    mods["AgentEval"] = importlib.import_module("WindGym.AgentEval")
    return mods


class ScriptedModel:
    def __init__(self, actions):
        self.actions, self.k = actions, 0

    def predict(self, obs, deterministic=False):
        a = self.actions[self.k]
        self.k += 1
        return a, None


def run(mods, name, cfg, two, ws, wd, ti, t_sim, seed):
    N = cfg["farm"]["nx"] * cfg["farm"]["ny"]
    rng = np.random.default_rng(seed)
    mg.SCRIPTS.clear()
    mg.SCRIPTS[0] = mg.draw_script(rng, 600, N)
    mg.SCRIPTS[1] = mg.draw_script(rng, 600, N)
    mg.SCRIPTS["two_farms"] = two
    mg.SCRIPTS["cursor"] = {0: 0, 1: 0}
    mg._FS_COUNT[0] = 0
    with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
        yaml.safe_dump(cfg, f)
        ypath = f.name
    env = mods["FarmEval"].FarmEval(turbine=mg.FakeTurbine(), yaml_path=ypath, turbtype="None", Baseline_comp=two,
                                    yaw_init="Zeros", seed=seed, reset_init=False)
    actions = rng.uniform(-1, 1, size=(t_sim + 2, N)).astype(np.float32)
    ds = mods["AgentEval"].eval_single_fast(env, ScriptedModel(actions), model_step=7, ws=ws, ti=ti, wd=wd,
                                            turbbox="Default", t_sim=t_sim)
    os.unlink(ypath)
    out = {}
    dims = {}
    for k, (d, arr) in ds.data_vars.items():
        out["var__" + k] = np.asarray(arr)
        dims[k] = list(d)
    for k, v in ds.coords.items():
        out["coord__" + k] = np.asarray(v)
    out["actions"] = actions
    for fi in (0, 1):
        n = mg.SCRIPTS["cursor"][fi] + 2
        out[f"script{fi}_uvw"] = mg.SCRIPTS[fi]["uvw"][:n]
        out[f"script{fi}_power"] = mg.SCRIPTS[fi]["power"][:n]
    out["meta"] = np.array(json.dumps(dict(name=name, cfg=cfg, two_farms=bool(two), ws=ws, wd=wd, ti=ti, t_sim=t_sim,
                                            seed=seed, dims=dims, var_order=list(ds.data_vars.keys()),
                                            coord_order=list(ds.coords.keys()))))
    path = os.path.join(HERE, f"eval_single_{name}.npz")
    np.savez_compressed(path, **out)
    print(name, {k: v.shape for k, v in out.items() if k.startswith("var__")}, os.path.getsize(path))


def main():
    mods = import_eval()
    run(mods, "baseline", mg.base_cfg(), True, 9.5, 265.0, 0.07, 40, 11)
    run(mods, "single_farm", mg.base_cfg(power_def=dict(Power_reward="Power_avg", Power_avg=10, Power_scaling=1.0),
                                         ActionMethod="yaw"), False, 12.0, 278.0, 0.04, 30, 12)


if __name__ == "__main__":
    main()
