"""External anchors of the flow physics: the only DYNAMIKS-produced numbers in the reference tree
(tests/golden/dynamiks_anchors.json, extracted by tests/golden/make_dynamiks_anchors.py from
examples/"Example 1 Make environment.ipynb" cell 4 and examples/PPO_2975000.zip -> _last_obs).

They are NOT parity fixtures — the wind conditions behind the PPO observations are unrecorded and the runs are
turbulent — but they are real outputs of the reference's physics (DYNAMIKS @77f4f87: jDWM Ainslie deficit, Hill-vortex
particle motion, Mann-box inflow) for Env1.yaml, and model M0 has to be plausible against them.  Stated bands:

 (A) PPO `_last_obs`: 32 samples of r = (25-step mean rotor wind speed of the downstream turbine) / (upstream turbine)
     at 8 D spacing, each with the yaws the trained agent held.  M0 is run with those yaws held and the unknown
     (ws, wd, TI) drawn from the Env1.yaml ranges, inflow = frozen Mann box with DWM meandering.  The pooled distributions
     must agree: |median| <= 0.04, |10th / 90th percentile| <= 0.07, and >= 80 % of the reference samples must fall inside
     the central 95 % band M0 predicts for their env (measured: median 0.963 vs 0.955, p10 0.817 vs 0.825, p90 1.105 vs
     1.100, 30 of 32 inside).
 (B) notebook: two identically placed downstream turbines show 25-step means of 0.915 and 0.522 of their upstream
     neighbours at TI = 2.7 %.  M0 (momentum-conserving Gaussian profile, as north_star asks) predicts 0.77 with a
     central 95 % band of about [0.69, 0.85] under Mann inflow: the band lies INSIDE the reference's range and contains
     the mean of the two values (0.72), but neither individual value — DWM's Ainslie deficit at this low TI is deeper and
     narrower than any momentum-conserving Gaussian can be (a fully immersed rotor cannot see less than ~0.65 with a
     Gaussian; the reference shows 0.52).  That is the physics gap that stays unpinned.
"""
import json
import os

import numpy as np
import pytest

from helpers import GOLDEN
from windgym_amd.config import EnvConfig
from windgym_amd.presets import env1_config
from windgym_amd.turbine import V80

A = json.load(open(os.path.join(GOLDEN, "dynamiks_anchors.json")))
WS_REF = np.array(A["ppo"]["ws"])
YAW_REF = np.array(A["ppo"]["yaw"])
R_REF = np.stack([WS_REF[:, 1] / WS_REF[:, 0], WS_REF[:, 3] / WS_REF[:, 2]], axis=1)      # [16 envs, 2 pairs]
BOX_SPEC = dict(dims=(1024, 256, 32), spacing=(3.0, 3.0, 3.0), seed=1234)


@pytest.fixture(scope="module")
def mann_box():
    from windgym_amd.mann import generate_mann_box
    return generate_mann_box(BOX_SPEC["dims"], BOX_SPEC["spacing"], seed=BOX_SPEC["seed"])


def _cfg(yaw, K, wind=None):
    d = env1_config()
    d["ActionMethod"] = "wind"                     # Env1.yaml's action method: the action is the yaw set-point
    if wind:
        d["wind"].update(wind)
    return EnvConfig(turbine=V80(), yaml_dict=d, turbtype="MannGenerate", n_envs=K, autoreset=False, n_passthrough=50,
                     n_rotor_pts=16, yaw_init="Defined", yaw_defined=list(yaw))


def _ratios(make, box, yaw, K, seed0, steps=45, wind=None):
    """25-step mean rotor wind speed ratios [K, 2] with the yaws held (the Env1.yaml observation, un-scaled)."""
    cfg = _cfg(yaw, K, wind)
    env = make(cfg)
    env.set_turbulence_box(box, BOX_SPEC["spacing"])
    obs = env.reset(seeds=seed0 + np.arange(K))
    a = np.tile((2 * (np.asarray(yaw) + 45) / 90 - 1)[None].astype(np.float32), (K, 1))
    for _ in range(steps):
        obs = env.step(a)[0]
    obs = np.asarray(obs.cpu().numpy() if hasattr(obs, "cpu") else obs)
    ws = (obs[:, 0::2] + 1) / 2 * 23 + 2
    return np.stack([ws[:, 1] / ws[:, 0], ws[:, 3] / ws[:, 2]], axis=1)


def _check_ppo(model):      # model [16, K, 2]
    med_m, med_r = np.median(model), np.median(R_REF)
    assert abs(med_m - med_r) <= 0.04, (med_m, med_r)
    for q in (10, 90):
        assert abs(np.percentile(model, q) - np.percentile(R_REF, q)) <= 0.07, (q, np.percentile(model, q), np.percentile(R_REF, q))
    lo, hi = np.percentile(model, 2.5, axis=1), np.percentile(model, 97.5, axis=1)
    inside = (R_REF >= lo) & (R_REF <= hi)
    assert inside.mean() >= 0.80, inside.sum()
    # the deepest wake the reference shows (0.59) is within what M0 produces at these spacings
    assert model.min() <= R_REF.min() + 0.05


def test_ppo_last_obs_distribution_oracle(oracle_lib, mann_box):
    K = 24
    model = np.array([_ratios(oracle_lib.Oracle, mann_box, YAW_REF[e], K, 1000 * e) for e in range(16)])
    _check_ppo(model)


def _notebook_band(make, box, K):
    nb = A["notebook"]
    wind = dict(ws_min=nb["Wind speed Global"], ws_max=nb["Wind speed Global"], TI_min=nb["Turbulence intensity"],
                TI_max=nb["Turbulence intensity"], wd_min=nb["Wind direction Global"], wd_max=nb["Wind direction Global"])
    r = _ratios(make, box, nb["yaw angles agent"], K, 0, wind=wind).ravel()
    m = np.array(nb["Wind speed at turbines measured"])
    return r, (m[1] / m[0], m[3] / m[2])


def test_notebook_wake_depth_band_oracle(oracle_lib, mann_box):
    r, (ref_a, ref_b) = _notebook_band(oracle_lib.Oracle, mann_box, 128)
    lo, hi = np.percentile(r, 2.5), np.percentile(r, 97.5)
    assert 0.51 < ref_b < 0.53 and 0.90 < ref_a < 0.93                    # the reference's two 25-step means
    assert ref_b < lo and hi < ref_a                                      # M0's band lies inside the reference's range
    assert lo < 0.5 * (ref_a + ref_b) < hi                                # ... and contains their mean
    assert 0.74 < np.median(r) < 0.80                                     # M0's own value, pinned


@pytest.mark.gpu
def test_ppo_last_obs_distribution_hip(mann_box):
    import torch
    from windgym_amd import binding
    assert torch.cuda.is_available()

    class _Hip(binding.HipBatch):
        def step(self, a):
            return super().step(torch.as_tensor(a, device="cuda"))

    K = 96
    model = np.array([_ratios(_Hip, mann_box, YAW_REF[e], K, 1000 * e) for e in range(16)])
    _check_ppo(model)
    r, (ref_a, ref_b) = _notebook_band(_Hip, mann_box, 512)
    assert ref_b < np.percentile(r, 2.5) and np.percentile(r, 97.5) < ref_a
