"""Deficit option "ainslie" (VERDICT r3 item 7): the eddy-viscosity deficit of the DWM model — what the reference's
``particleDeficitGenerator=jDWMAinslieGenerator()`` (WindGym/Wind_Farm_Env.py:706, :774) solves inside DYNAMIKS — solved on
the host (windgym_amd/ainslie.py) and sampled from a 4-D table by the flow kernels.  The solver is checked against what
the equations guarantee (momentum-deficit flux, the constant-viscosity limit), the oracle's lookup against a direct numpy
evaluation of the table, the HIP path against the oracle; the DYNAMIKS notebook anchor is re-run with it."""
import numpy as np
import pytest

from windgym_amd import ainslie, presets
from windgym_amd.config import EnvConfig, rotor_points
from windgym_amd.turbine import V80


def _cfg(deficit="ainslie", n_envs=1, turbtype="None", ws=9.3, ti=0.06, nx=2, ny=1, **kw):
    d = presets.env1_config()
    d["farm"].update(nx=nx, ny=ny, xDist=6, yDist=4)
    d["wind"] = dict(ws_min=ws, ws_max=ws, wd_min=270, wd_max=270, TI_min=ti, TI_max=ti)
    d["yaw_init"] = "Zeros"
    d["ActionMethod"] = "yaw"
    return EnvConfig(turbine=V80(), yaml_dict=d, turbtype=turbtype, n_envs=n_envs, autoreset=False, n_rotor_pts=16,
                     deficit=deficit, **kw)


def test_config_switch():
    c = _cfg().to_c()
    g = _cfg("gaussian").to_c()
    assert c.deficit_model == 2 and (c.m0_ka, c.m0_kb, c.m0_eps) == (g.m0_ka, g.m0_kb, g.m0_eps)


def test_solver_conserves_the_momentum_deficit_flux():
    """U dU/dx + V dU/dr = (1/r) d/dr(nu r dU/dr) with continuity conserves int U (1 - U) r dr whatever nu(x) is."""
    xs = np.array([0.0, 2.0, 6.0, 16.0, 40.0])
    r, d = ainslie.solve(np.array([0.3, 0.8, 0.94])[:, None], np.array([0.02, 0.12])[None, :], xs)
    flux = np.sum((1.0 - d) * d * r, axis=-1)
    np.testing.assert_allclose(flux, np.broadcast_to(flux[..., :1], flux.shape), rtol=0.04)   # (non-conservative form: a few % at Ct 0.94)
    assert (np.diff(d[..., 0], axis=-1) <= 1e-9).all()                  # the centre-line deficit only decays
    assert d[0, 1, -1, 0] < d[0, 0, -1, 0] and d[1, 1, -1, 0] < d[1, 0, -1, 0]      # faster in higher ambient TI


def test_solver_constant_viscosity_limit_is_the_spreading_gaussian():
    """Small deficit, constant nu: the equation linearises to radial diffusion — a top hat of depth 2a and radius r_w
    tends to the Gaussian of variance r_w^2 / 4 + 2 nu x whose centre deficit is a r_w^2 / (r_w^2 / 4 + 2 nu x)."""
    ct, nu = 0.04, 0.02
    a = 0.5 * (1 - np.sqrt(1 - ct))
    rw = np.sqrt((1 - a) / (1 - 2 * a)) * (1 - 0.45 * a * a)
    xs = np.array([0.0, 30.0, 60.0])
    r, d = ainslie.solve(ct, 0.1, xs, r_max=8.0, nr=320, nu_const=nu)
    want = a * rw ** 2 / (rw ** 2 / 4 + 2 * nu * xs[1:])
    np.testing.assert_allclose(d[1:, 0], want, rtol=0.03)
    s2 = rw ** 2 / 4 + 2 * nu * xs[2]
    np.testing.assert_allclose(d[2, :120] / d[2, 0], np.exp(-r[:120] ** 2 / (2 * s2)), atol=0.02)


def test_table_shape_and_rotor_boundary_condition():
    tab, s = ainslie.deficit_table()
    assert tab.shape == (len(s["ct"]), len(s["ti"]), s["n_x"], s["n_r"]) and tab.dtype == np.float32
    assert np.isfinite(tab).all() and tab.min() >= 0.0
    # x = 0, r = 0: the deficit is 2a of the uniformly loaded rotor
    np.testing.assert_allclose(tab[:, :, 0, 0], np.broadcast_to((1 - np.sqrt(1 - s["ct"]))[:, None], tab.shape[:2]), rtol=1e-6)
    assert tab[..., -1].max() < 0.01                                     # the table's outer radius sits outside every wake


def _table_lookup(tab, s, ct, ti, xd, r_over_R):
    """Independent numpy restatement of the 4-linear sampling rule of include/windgym_hip.h (wg_set_deficit_table)."""
    def ax(f, n):
        f = min(max(f, 0.0), n - 1.0)
        i = min(int(f), n - 2)
        return i, f - i
    ic, wc = ax((ct - s["ct"][0]) / (s["ct"][-1] - s["ct"][0]) * (len(s["ct"]) - 1), len(s["ct"]))
    it, wt = ax(np.log(ti / s["ti"][0]) / np.log(s["ti"][-1] / s["ti"][0]) * (len(s["ti"]) - 1), len(s["ti"]))
    ix, wx = ax(xd / s["x_max_D"] * (s["n_x"] - 1), s["n_x"])
    prof = sum((wc if a else 1 - wc) * (wt if b else 1 - wt) * (wx if c else 1 - wx) * tab[ic + a, it + b, ix + c].astype(np.float64)
               for a in (0, 1) for b in (0, 1) for c in (0, 1))
    fr = np.asarray(r_over_R) / s["r_max_R"] * (s["n_r"] - 1)
    out = np.zeros_like(fr)
    ok = fr < s["n_r"] - 1.5                      # (the last half cell is cut: the slope stencil needs node m + 1)
    ir = fr[ok].astype(int)
    out[ok] = prof[ir] + (fr[ok] - ir) * (prof[ir + 1] - prof[ir])
    return out


@pytest.mark.parametrize("ws,ti", [(9.3, 0.06), (7.0, 0.027), (12.5, 0.15)])
def test_oracle_steady_single_wake_equals_the_table(ws, ti, oracle_lib):
    """Two turbines in line, zero yaw, steady inflow, no TI folding: once the chain has developed the downstream rotor wind
    is U (1 - mean_p table(Ct, TI, x / D, r_p / R))."""
    cfg = _cfg(ws=ws, ti=ti, wake_ti_fold=False)
    o = oracle_lib.Oracle(cfg)
    o.reset(seeds=[1])
    a = np.zeros((1, 2), np.float32)
    for _ in range(200):
        o.step(a)
    u = o.info("rotor_uvw_agent")[0, :, 0]
    tabt = cfg.tab
    ct = float(np.interp(ws, tabt.ws_tab, tabt.ct_tab))
    x = o.info("turb_x")[0]
    dy, dz = rotor_points(16, 40.0)
    tab, s = ainslie.deficit_table()
    want = ws * (1 - np.mean(_table_lookup(tab, s, ct, ti, float(x[1] - x[0]) / 80.0, np.hypot(dy, dz) / 40.0)))
    assert abs(u[0] - ws) < 1e-12
    np.testing.assert_allclose(u[1], want, rtol=1e-6)                   # (Ct, k carried as float records along the chain)
    assert u[1] < ws * 0.95


def test_ainslie_wake_is_deeper_and_narrower_than_the_gaussian_at_low_ti(oracle_lib):
    """At TI = 2.7 % (the notebook's) and 6 D the eddy-viscosity deficit has barely left the near wake:
    the rotor-averaged deficit is larger than the Gaussian's, the profile still carries its top-hat shoulders."""
    res = {}
    for dm in ("gaussian", "ainslie"):
        d = _cfg(dm, ws=8.0, ti=0.027, wake_ti_fold=False)
        o = oracle_lib.Oracle(d)
        o.reset(seeds=[1])
        for _ in range(200):
            o.step(np.zeros((1, 2), np.float32))
        res[dm] = o.info("rotor_uvw_agent")[0, 1, 0] / 8.0
    assert res["ainslie"] < res["gaussian"] - 0.03                       # measured: 0.713 vs 0.761 of the free wind


@pytest.mark.gpu
def test_uniform_ring_variant_refuses_the_option():
    import os
    from windgym_amd import binding
    os.environ["WG_FLOW_BLOCK"] = "256"              # uniform rings / sample-major phases
    try:
        with pytest.raises(NotImplementedError):
            binding.HipBatch(_cfg(n_envs=2, nx=3, ny=3))
    finally:
        del os.environ["WG_FLOW_BLOCK"]


@pytest.mark.gpu
@pytest.mark.parametrize("block,turbtype", [(64, "None"), (128, "None"), (256, "None"), (64, "MannFixed"), (64, "Random")])
def test_hip_matches_oracle_ainslie(block, turbtype, oracle_lib):
    import os
    import torch
    from windgym_amd import binding
    from windgym_amd.mann import generate_mann_box
    nxy = 6 if block == 256 else 3                   # 36 turbines: the compact variant at 256 threads (large steady farms)
    cfg = _cfg(n_envs=3, turbtype=turbtype, nx=nxy, ny=nxy, ws=10.0, ti=0.08)
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


@pytest.mark.gpu
def test_reset_without_a_table_is_refused():
    from windgym_amd import binding
    env = binding.HipBatch(_cfg(n_envs=2))
    env._dtab = object()                              # keep the binding from installing the default table
    with pytest.raises(Exception, match="wg_set_deficit_table"):
        env.reset(seeds=[1, 2])


def test_notebook_anchor_with_the_eddy_viscosity_deficit(oracle_lib):
    """The one hard DYNAMIKS number in the reference tree (notebook cell 4: downstream / upstream 25-step means 0.915 and
    0.522 at TI = 2.7 %, 8 D; tests/test_dynamiks_anchors.py) against this option.  Measured (oracle, Mann inflow, 96
    draws): median 0.744, 95 % band [0.666, 0.848], minimum 0.639 (Gaussian: 0.780, [0.716, 0.867], 0.697) — the deficit
    the reference's own model family gives is deeper, and a centred rotor sees 0.606 of the free wind at Ct 0.8 (table,
    below) where no momentum-conserving Gaussian goes under 0.65; the reference's 0.522 still lies outside: the published
    DWM calibration restated here is not DYNAMIKS'  — physics parity stays unpinned."""
    import test_dynamiks_anchors as T
    from windgym_amd.mann import generate_mann_box
    box = generate_mann_box(T.BOX_SPEC["dims"], T.BOX_SPEC["spacing"], seed=T.BOX_SPEC["seed"])
    orig = T._cfg

    def cfgf(yaw, K, wind=None):
        c = orig(yaw, K, wind)
        c.deficit = "ainslie"
        return c
    T._cfg = cfgf
    try:
        r, (ref_a, ref_b) = T._notebook_band(oracle_lib.Oracle, box, 48)
    finally:
        T._cfg = orig
    assert 0.71 < np.median(r) < 0.775 and r.min() < 0.68
    assert ref_b < np.percentile(r, 2.5) and np.percentile(r, 97.5) < ref_a
    tab, s = ainslie.deficit_table()
    dy, dz = rotor_points(16, 40.0)
    centred = 1 - np.mean(_table_lookup(tab, s, 0.8, 0.027, 8.0, np.hypot(dy, dz) / 40.0))
    assert 0.58 < centred < 0.63


@pytest.mark.gpu
@pytest.mark.parametrize("deficit", ["ainslie", "super_gaussian"])
def test_flow_field_view_follows_the_deficit_option(deficit, oracle_lib):
    """wg_get_windspeed (fs.get_windspeed(XYView(...)), Wind_Farm_Env.py:1040-1083) draws the wakes with the deficit model
    the handle was created with."""
    import torch
    from windgym_amd import binding
    cfg = _cfg(deficit, n_envs=2, nx=3, ny=2, ws=10.0, ti=0.07, advect_full_chains=True)
    cfg.wd_min, cfg.wd_max = 262.0, 278.0
    env, orc = binding.HipBatch(cfg), oracle_lib.Oracle(cfg)
    env.reset(seeds=[5, 6]), orc.reset(seeds=[5, 6])
    rng = np.random.default_rng(3)
    for _ in range(60):
        a = rng.uniform(-1, 1, size=(2, cfg.n_turb)).astype(np.float32)
        env.step(torch.as_tensor(a, device="cuda")), orc.step(a)
    tx, ty = orc.info("turb_x"), orc.info("turb_y")
    for b in (0, 1):
        xs = np.linspace(tx[b].min() - 100.0, tx[b].max() + 800.0, 83).astype(np.float32)
        ys = np.linspace(ty[b].min() - 200.0, ty[b].max() + 200.0, 57).astype(np.float32)
        got = env.windspeed(b, xs, ys).cpu().numpy()
        np.testing.assert_allclose(got, orc.windspeed(b, xs, ys), rtol=2e-4, atol=2e-3)
        assert (10.0 - got[0]).max() > 2.0


@pytest.mark.gpu
def test_gym_env_accepts_the_option(tmp_path):
    """`WindFarmEnv(..., deficit="ainslie")`: the Gymnasium facade hands the option to the config, the binding installs the
    table at the first reset; a waked turbine produces less than with the Gaussian profile at low TI."""
    import yaml
    import windgym_amd as wg
    d = presets.env1_config()
    d["farm"].update(nx=2, ny=1, xDist=6, yDist=4)
    d["wind"] = dict(ws_min=8.0, ws_max=8.0, wd_min=270, wd_max=270, TI_min=0.03, TI_max=0.03)
    d["yaw_init"] = "Zeros"
    p = tmp_path / "farm.yaml"
    p.write_text(yaml.safe_dump(d))
    powers = {}
    for dm in ("gaussian", "ainslie"):
        env = wg.WindFarmEnv(turbine=wg.V80(), n_passthrough=2, yaml_path=str(p), turbtype="None", seed=3, deficit=dm)
        env.reset(seed=3)
        for _ in range(30):
            obs, rew, term, trunc, info = env.step(np.zeros(env.action_space.shape, dtype=np.float32))
        powers[dm] = np.asarray(info["Power pr turbine agent"] if "Power pr turbine agent" in info else info["Power agent"], dtype=np.float64)
        env.close()
    assert np.sum(powers["ainslie"]) < np.sum(powers["gaussian"]) * 0.99
