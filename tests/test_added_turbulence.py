"""Row a7: wake-added small-scale turbulence (reference: addedTurbulenceModel = [Synchronized]AutoScalingIsotropicMann-
Turbulence(), Wind_Farm_Env.py:618, :638, :644, :659).  The oracle restates the DWM scaling k_mt = km1 |dU| + km2 |d dU/dr|
of an isotropic unit-variance field; the HIP path must match it in every turbulent inflow mode and kernel variant, and
the `wd` / TI sensors of WAKED turbines must respond to it while free-stream turbines do not."""
import numpy as np
import pytest

from oracle import oracle as om
from windgym_amd import presets
from windgym_amd.config import EnvConfig
from windgym_amd.mann import generate_mann_box
from windgym_amd.turbine import V80


def _cfg(turbtype, added, n_envs=2, **kw):
    d = presets.bench_cfg2_config()
    d["wind"] = dict(d.get("wind", {}), ws_min=9, ws_max=9, wd_min=270, wd_max=270, TI_min=0.06, TI_max=0.06)
    return EnvConfig(turbine=V80(), yaml_dict=d, turbtype=turbtype, n_envs=n_envs, autoreset=False, n_passthrough=5,
                     n_rotor_pts=16, added_turbulence=added, **kw)


def _rollout(env, n_steps, n_act):
    uvw = []
    a = np.zeros((env.B, n_act), np.float32)
    for _ in range(n_steps):
        env.step(a)
        uvw.append(env.info("rotor_uvw_agent").copy())
    return np.array(uvw)


@pytest.fixture(scope="module")
def small_box():
    return generate_mann_box((256, 64, 32), (3.0, 3.0, 3.0), seed=5), (3.0, 3.0, 3.0)


def test_config_auto_follows_the_reference():
    assert _cfg("None", "auto").to_c().added_turbulence == 0          # :664 addedTurbulenceModel = None
    for tt in ("Random", "MannFixed", "MannGenerate"):
        assert _cfg(tt, "auto").to_c().added_turbulence == 1          # :618, :638, :644, :659
        assert _cfg(tt, "none").to_c().added_turbulence == 0
    with pytest.raises(ValueError):
        _cfg("Random", "bogus").to_c()


def test_oracle_waked_turbines_feel_it_and_free_stream_ones_do_not(small_box):
    res = {}
    for added in ("none", "iso"):
        o = om.Oracle(_cfg("MannFixed", added))
        o.set_turbulence_box(*small_box)
        o.reset(seeds=[1, 2])
        res[added] = _rollout(o, 150, 16)
    d = res["iso"] - res["none"]                       # [steps, B, N, 3]
    # wd = 270: the 4x4 grid's first column (lowest flow-frame x) is unwaked
    x = o.info("turb_x")[0]
    front = np.argsort(x)[:4]
    back = np.setdiff1d(np.arange(16), front)
    assert np.abs(d[:, :, front]).max() < 1e-9
    # every component of the waked rotors' inflow fluctuates more: the field is isotropic
    for cc in range(3):
        assert d[60:, :, back, cc].std() > 0.02
    # the lateral component is what the wd sensor sees (arctan(v / u))
    assert res["iso"][60:, :, back, 1].std() > res["none"][60:, :, back, 1].std()


def test_oracle_ti_fold_switch(small_box):
    """wake_ti_fold=False: the Crespo-Hernandez TI is not folded into the emitted particles -> deep-array wakes recover
    more slowly (narrower wakes, larger centre-line deficit)."""
    u = {}
    for fold in (True, False):
        o = om.Oracle(_cfg("None", "none", n_envs=1, wake_ti_fold=fold))
        o.reset(seeds=[3])
        u[fold] = _rollout(o, 250, 16)[-1, 0, :, 0]
    x = o.info("turb_x")[0]
    last = np.argsort(x)[-4:]
    assert np.all(u[False][last] < u[True][last] - 1e-3)


# (k_flow's instantiations under both turbulent inflows; "envb" / "envb1" = k_flow_envb, the one-launch frozen-box kernel, with two
# waves / one wave per env ("envb4": four, one per farm slot) — it serves the box inflow only)
HIP_CASES = [(t, b) for t in ("MannFixed", "Random") for b in (64, 128, 256)] + [("MannFixed", "envb4"), ("MannFixed", "envb"), ("MannFixed", "envb1")]


@pytest.mark.gpu
@pytest.mark.parametrize("turbtype,block", HIP_CASES)
def test_hip_matches_oracle_with_added_turbulence(turbtype, block, small_box):
    import os
    import torch
    from windgym_amd import binding
    cfg = _cfg(turbtype, "iso", n_envs=3)
    envk = isinstance(block, str)
    hooks = {"WG_FLOW_BLOCK": "64" if envk else str(block), "WG_FLOW_ENV": "1" if envk else "0"}
    if envk:
        hooks["WG_ENV_WPE"] = block[-1] if block[-1] in "14" else "2"
    os.environ.update(hooks)
    try:
        env = binding.HipBatch(cfg)
    finally:
        for k in hooks:
            del os.environ[k]
    assert env.flow_variant()[0] == (64 if envk else block) and env.flow_variant()[2] == (2 if envk else 0)
    o = om.Oracle(cfg)
    if turbtype != "Random":
        env.set_turbulence_box(*small_box)
        o.set_turbulence_box(*small_box)
    seeds = [11, 12, 13]
    env.reset(seeds=seeds)
    o.reset(seeds=seeds)
    rng = np.random.default_rng(0)
    for i in range(80):
        a = rng.uniform(-1, 1, size=(3, 16)).astype(np.float32)
        obs, rew, _, _ = env.step(torch.as_tensor(a, device="cuda"))
        oo, orew, _, _ = o.step(a)
        np.testing.assert_allclose(obs.cpu().numpy(), oo, atol=2e-4, rtol=0)
        uvw_h = env.info("rotor_uvw_agent").cpu().numpy()
        uvw_o = o.info("rotor_uvw_agent")
        # rel 1e-4 of the wind speed on every component (the added v, w are ~0.1 m/s: absolute tolerance)
        np.testing.assert_allclose(uvw_h, uvw_o, atol=1.5e-3, rtol=1e-4)
    env.check()


@pytest.mark.gpu
def test_hip_wd_sensor_of_waked_turbines_responds(small_box):
    import torch
    from windgym_amd import binding
    wd = {}
    for added in ("none", "iso"):
        env = binding.HipBatch(_cfg("MannFixed", added, n_envs=4))
        env.set_turbulence_box(*small_box)
        env.reset(seeds=[1, 2, 3, 4])
        a = torch.zeros((4, 16), device="cuda")
        rec = []
        for _ in range(150):
            env.step(a)
            rec.append(env.info("wd_turb").cpu().numpy().copy())
        wd[added] = np.array(rec)[60:]
        x = env.info("turb_x").cpu().numpy()[0]
    front = np.argsort(x)[:4]
    back = np.setdiff1d(np.arange(16), front)
    np.testing.assert_allclose(wd["iso"][:, :, front], wd["none"][:, :, front], atol=1e-4)
    assert wd["iso"][:, :, back].std() > 1.02 * wd["none"][:, :, back].std()
