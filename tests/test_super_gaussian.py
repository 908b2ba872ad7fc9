"""Deficit option "super_gaussian" (VERDICT r2 item 8): Blondel & Cathelain (2020) — the wake model of the reference's
PyWakeAgent — as a switch of model M0 (default stays the Gaussian north_star asks for).  Oracle pinned against the
published formula evaluated directly; HIP path against the oracle; the DYNAMIKS notebook anchor re-run with it."""
import math

import numpy as np
import pytest

from oracle import oracle as om
from windgym_amd import presets
from windgym_amd.config import EnvConfig
from windgym_amd.turbine import V80


def _cfg(deficit, n_envs=1, turbtype="None", ws=9.3, ti=0.06, nx=2, ny=1, **kw):
    d = presets.env1_config()
    d["farm"].update(nx=nx, ny=ny, xDist=6, yDist=4)
    d["wind"] = dict(ws_min=ws, ws_max=ws, wd_min=270, wd_max=270, TI_min=ti, TI_max=ti)
    d["yaw_init"] = "Zeros"
    d["ActionMethod"] = "yaw"
    return EnvConfig(turbine=V80(), yaml_dict=d, turbtype=turbtype, n_envs=n_envs, autoreset=False, n_rotor_pts=16,
                     deficit=deficit, **kw)


def test_config_switch_and_default_constants():
    c = _cfg("super_gaussian").to_c()
    assert c.deficit_model == 1 and (c.m0_ka, c.m0_kb, c.m0_eps) == (0.17, 0.005, 0.2)     # Blondel & Cathelain Table 2
    assert _cfg("gaussian").to_c().deficit_model == 0
    c2 = _cfg("super_gaussian", model_constants=dict(ka=0.2)).to_c()
    assert c2.m0_ka == 0.2 and c2.m0_kb == 0.005
    with pytest.raises(ValueError):
        _cfg("top_hat").to_c()


def test_oracle_steady_single_wake_equals_the_published_formula(oracle_lib):
    """Two turbines 6 D apart, zero yaw, steady inflow, no TI folding: once the chain has developed, the downstream
    rotor wind is U (1 - mean_p C exp(-(r_p/D)^n / (2 sigma~^2))) with n, sigma~, C of Blondel & Cathelain (2020)."""
    ws, ti = 9.3, 0.06
    cfg = _cfg("super_gaussian", ws=ws, ti=ti, wake_ti_fold=False)
    o = oracle_lib.Oracle(cfg)
    o.reset(seeds=[1])
    a = np.zeros((1, 2), np.float32)
    for _ in range(200):
        o.step(a)
    u = o.info("rotor_uvw_agent")[0, :, 0]
    tab = cfg.tab
    ct = float(np.interp(ws, tab.ws_tab, tab.ct_tab))
    x = o.info("turb_x")[0]
    xd = float(x[1] - x[0]) / 80.0                    # (the reference's linspace layout rule: not simply xDist)
    beta = 0.5 * (1 + math.sqrt(1 - ct)) / math.sqrt(1 - ct)
    sig = (0.17 * ti + 0.005) * xd + 0.2 * math.sqrt(beta)
    n = 3.11 * math.exp(-0.68 * xd) + 2.41
    C = 2 ** (2 / n - 1) - math.sqrt(2 ** (4 / n - 2) - n * ct / (16 * math.gamma(2 / n) * sig ** (4 / n)))
    from windgym_amd.config import rotor_points
    dy, dz = rotor_points(16, 40.0)
    r = np.hypot(dy, dz) / 80.0
    want = ws * (1 - np.mean(C * np.exp(-r ** n / (2 * sig ** 2))))
    assert abs(u[0] - ws) < 1e-12
    np.testing.assert_allclose(u[1], want, rtol=1e-9)
    # deeper and flatter than the Gaussian at the same width parameters
    og = oracle_lib.Oracle(_cfg("gaussian", ws=ws, ti=ti, wake_ti_fold=False, model_constants=dict(ka=0.17, kb=0.005, eps=0.2)))
    og.reset(seeds=[1])
    for _ in range(200):
        og.step(a)
    assert u[1] < og.info("rotor_uvw_agent")[0, 1, 0]


@pytest.mark.gpu
def test_uniform_ring_variant_refuses_the_option():
    import os
    from windgym_amd import binding
    os.environ["WG_FLOW_BLOCK"] = "256"              # uniform rings / sample-major phases
    try:
        with pytest.raises(NotImplementedError):
            binding.HipBatch(_cfg("super_gaussian", n_envs=2, nx=3, ny=3))
    finally:
        del os.environ["WG_FLOW_BLOCK"]


@pytest.mark.gpu
@pytest.mark.parametrize("block,turbtype", [(64, "None"), (128, "None"), (256, "None"), (64, "MannFixed"), (64, "Random")])
def test_hip_matches_oracle_super_gaussian(block, turbtype, oracle_lib):
    import os
    import torch
    from windgym_amd import binding
    from windgym_amd.mann import generate_mann_box
    nxy = 6 if block == 256 else 3                   # 36 turbines: the compact variant at 256 threads (large steady farms)
    cfg = _cfg("super_gaussian", n_envs=3, turbtype=turbtype, nx=nxy, ny=nxy, ws=10.0, ti=0.08)
    cfg.wd_min, cfg.wd_max = 255.0, 285.0
    if block != 256:
        os.environ["WG_FLOW_BLOCK"] = str(block)
    try:
        env = binding.HipBatch(cfg)
    finally:
        os.environ.pop("WG_FLOW_BLOCK", None)
    assert env.flow_variant() == (block, True, False)
    o = oracle_lib.Oracle(cfg)
    if turbtype == "MannFixed":
        box = generate_mann_box((256, 64, 32), (3.0, 3.0, 3.0), seed=9)
        env.set_turbulence_box(box, (3.0, 3.0, 3.0)), o.set_turbulence_box(box, (3.0, 3.0, 3.0))
    seeds = [21, 22, 23]
    np.testing.assert_allclose(env.reset(seeds=seeds).cpu().numpy(), o.reset(seeds=seeds), rtol=0, atol=3e-4)
    rng = np.random.default_rng(1)
    for i in range(100):
        a = rng.uniform(-1, 1, size=(3, cfg.n_turb)).astype(np.float32)
        obs, rew, _, _ = env.step(torch.as_tensor(a, device="cuda"))
        oo, orew, _, _ = o.step(a)
        np.testing.assert_allclose(obs.cpu().numpy(), oo, rtol=0, atol=3e-4, err_msg=f"step {i}")
        np.testing.assert_allclose(env.info("rotor_uvw_agent").cpu().numpy(), o.info("rotor_uvw_agent"), rtol=2e-4, atol=2e-3)
    env.check()


def test_notebook_anchor_with_the_super_gaussian(oracle_lib):
    """The one hard DYNAMIKS number in the reference tree (notebook cell 4: downstream / upstream 25-step means 0.915
    and 0.522 at TI = 2.7 %, tests/test_dynamiks_anchors.py) against BOTH deficit models.  Measured (oracle, Mann inflow,
    96 draws): Gaussian median 0.778, 95 % band [0.709, 0.866]; super-Gaussian median 0.746, band [0.667, 0.847] —
    deeper, as expected, but 0.522 stays outside: the spread between the reference's two identically placed turbines
    (0.52 vs 0.92 in the same 25 s) is meandering of a narrow Ainslie deficit, not a property of the mean profile."""
    import test_dynamiks_anchors as T
    from windgym_amd.mann import generate_mann_box
    box = generate_mann_box(T.BOX_SPEC["dims"], T.BOX_SPEC["spacing"], seed=T.BOX_SPEC["seed"])
    orig = T._cfg
    med = {}
    try:
        for dm in ("gaussian", "super_gaussian"):
            def cfgf(yaw, K, wind=None, dm=dm):
                c = orig(yaw, K, wind)
                c.deficit = dm
                return c
            T._cfg = cfgf
            r, (ref_a, ref_b) = T._notebook_band(oracle_lib.Oracle, box, 48)
            med[dm] = (np.median(r), np.percentile(r, 2.5), np.percentile(r, 97.5))
    finally:
        T._cfg = orig
    assert med["super_gaussian"][0] < med["gaussian"][0] - 0.015          # the super-Gaussian wake is deeper at 8 D, low TI
    assert 0.71 < med["super_gaussian"][0] < 0.78 and 0.74 < med["gaussian"][0] < 0.81
    for dm in med:                                                       # neither band reaches the reference's 0.522
        assert ref_b < med[dm][1] and med[dm][2] < ref_a
