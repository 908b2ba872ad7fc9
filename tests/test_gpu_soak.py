"""Full-size property soaks for BASELINE.json's cfg3, cfg4 and cfg5 (cfg2's lives in
test_gpu_parity.py) and the worst case of the 16-bit emission record.  No oracle at these sizes: the properties are the
ones the domain offers — finite, |obs| <= 1, episode counts == truncations, no sticky error (the background-developed
episode is always ready), every farm waked, >= 2 rollovers per env."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def hip():
    import torch
    from windgym_amd import binding
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    binding.load_library()
    return binding


def _soak(env, cfg, B, steps, min_roll, check_every=500):
    import torch
    g = torch.Generator(device="cpu").manual_seed(0)
    acts = (torch.rand((16, B, cfg.n_turb), generator=g) * 2 - 1).cuda()
    n_trunc = torch.zeros(B, dtype=torch.int64, device="cuda")
    for i in range(steps):
        obs, rew, tr, fin = env.step(acts[i % 16])
        n_trunc += tr.long()
        if i % check_every == 0:
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all() and (obs.abs() <= 1).all()
            assert torch.isfinite(fin).all()
    env.check()                                  # no NaN power, no unready background episode, no saturated record
    assert int(n_trunc.min()) >= min_roll, (int(n_trunc.min()), int(n_trunc.max()))
    np.testing.assert_array_equal(env.info("episode").cpu().numpy(), n_trunc.cpu().numpy())
    m = env.metrics().cpu().numpy()
    assert m[7] == steps * B and m[3] == int(n_trunc.sum())
    u = env.info("rotor_uvw_agent").cpu().numpy()[..., 0]
    return u, n_trunc


def test_soak_cfg2_headline_batch_4096_envs(hip):
    """The batch the headline metric is quoted on: bench.py's cfg2 configuration, 4096 envs x 16 turbines x
    2 farms on one GPU, default n_passthrough, 1200 steps (time_max <= 1121: every env rolls over at least once) — same
    assertions as the 1024-env soak of test_gpu_parity.py."""
    import bench
    B = 4096
    cfg = bench.make_cfg(B, workload="cfg2")
    env = hip.HipBatch(cfg)
    assert env.flow_variant() == (64, True, 2)              # k_flow_env, the kernel the bench line times
    env.reset(seeds=1234 + np.arange(B))
    u, n_trunc = _soak(env, cfg, B, 1200, 1, check_every=200)
    assert int(n_trunc.max()) <= 3
    tm = env.info("time_max").cpu().numpy()
    assert tm.min() >= 426 and tm.max() <= 1121              # int(5 * dist / ws), ws in [7, 15], 1280 m <= dist <= 1568 m
    ws = env.info("ws_global").cpu().numpy()
    # (among 4096 sampled wind directions a few leave every rotor of the 5.33 D grid outside the 5-sigma cut-off of the
    # upstream wakes: "waked" holds for the batch, not for every single env)
    assert (u.max(axis=1) <= ws * (1 + 1e-5)).all() and (u.min(axis=1) < ws - 0.02).mean() > 0.98


def test_soak_cfg3_horns_rev_512_envs(hip):
    """cfg3 at its per-GPU bench size: Horns Rev 1 (80 turbines) x 512 envs x 2 farms, compact / pair-major variant at
    256 threads with chunked targets, >= 2 rollovers per env (n_passthrough 1.5 keeps the run short: ~620-1330 steps per
    episode instead of 2100-4400)."""
    import bench
    B = 512
    cfg = bench.make_cfg(B, workload="cfg3")
    cfg.n_passthrough = 1.5
    env = hip.HipBatch(cfg)
    assert env.flow_variant() == (256, True, False)
    env.reset(seeds=1234 + np.arange(B))
    u, n_trunc = _soak(env, cfg, B, 2800, 2)
    ws = env.info("ws_global").cpu().numpy()
    assert (u.max(axis=1) <= ws * (1 + 1e-5)).all() and (u.min(axis=1) < ws - 0.05).all()      # every farm is waked
    assert int(n_trunc.max()) <= 6


def test_soak_cfg4_multi_agent_2048_envs_fused_buffer(hip):
    """cfg4: 3x3 farm, PettingZoo per-agent observations written by the step's own glue kernel into the registered
    buffer [B, 9, o_t + o_f], 2048 envs (k_flow_env, two waves per env), double timestep increment."""
    import torch
    import bench
    B = 2048
    cfg = bench.make_cfg(B, workload="cfg4")
    env = hip.HipBatch(cfg)
    assert env.flow_variant()[2] == 2            # k_flow_env: the env's slots in one / two waves
    multi = env.fuse_obs_multi()
    env.reset(seeds=1234 + np.arange(B))
    u, n_trunc = _soak(env, cfg, B, 1500, 2)
    assert torch.isfinite(multi).all() and (multi.abs() <= 1).all()
    assert (multi - env.obs_multi()).abs().max().item() <= 2e-6   # the fused buffer holds what wg_obs_multi returns (to float rounding: summation order)
    ws = env.info("ws_global").cpu().numpy()
    assert (u.max(axis=1) <= ws * (1 + 1e-5)).all()
    assert (u.min(axis=1) < ws - 0.02).mean() > 0.95       # (a 3x3 farm can be unwaked for a moment right after a rollover)


def test_soak_cfg5_reference_box_1024_envs(hip):
    """cfg5: 16 turbines x 1024 envs on the reference's 2048 x 512 x 64 box (0.8 GB, generated on the GPU), meandering,
    wake-added turbulence field on (reference default), >= 2 rollovers per env."""
    import torch
    import bench
    from windgym_amd.mann import generate_mann_box_torch, reference_box_spec
    B = 1024
    cfg = bench.make_cfg(B, workload="cfg5")
    env = hip.HipBatch(cfg)
    spec = reference_box_spec("MannFixed", cfg.D)
    env.set_turbulence_box(generate_mann_box_torch(device=torch.device("cuda"), **spec), spec["dxyz"])
    env.reset(seeds=1234 + np.arange(B))
    u, n_trunc = _soak(env, cfg, B, 2600, 2)
    assert int(n_trunc.max()) <= 7
    ws = env.info("ws_global").cpu().numpy()
    ti = env.info("ti_global").cpu().numpy()
    # turbulent inflow: rotor winds scatter around U by a few sigma = TI U; every farm still has a waked rotor
    assert (np.abs(u - ws[:, None]) < ws[:, None] * (0.6 + 6 * ti[:, None])).all()
    assert ((u.min(axis=1) < ws * (1 - 0.02))).mean() > 0.99


# ---------------------------------------------------------------------------------------------------------------------
# the 16-bit emission record at its worst case
# ---------------------------------------------------------------------------------------------------------------------
def _extreme_cfg(ti, ws, n_envs=2, **kw):
    from windgym_amd import presets
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    d = presets.bench_cfg2_config()
    d["wind"] = dict(ws_min=ws, ws_max=ws, wd_min=268, wd_max=272, TI_min=ti, TI_max=ti)
    d["farm"] = dict(d["farm"], yaw_min=-45, yaw_max=45)
    return EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None", n_envs=n_envs, autoreset=False, n_passthrough=5,
                     n_rotor_pts=16, **kw)


def test_u16_record_at_the_edge_of_its_range_matches_the_oracle(hip):
    """Largest values the config validation admits: TI = 0.5 (k = 0.38 sqrt(0.5^2 + added^2) + 0.004 up to 0.23 of the
    0.25 the unorm16 covers), 25 m/s with the yaws driven to +-45 deg (|hv| = 0.4 sin(45) 25 = 7.1 of 16 m/s; u_e as a
    16-bit fraction of 25 m/s: a step of 3.8e-4 m/s).  The quantisation steps (ct 1.5e-5, k 3.8e-6, u_e U / 65535, hv 4.9e-4 m/s)
    must keep the stated tolerances there too, and nothing saturates (wg_check)."""
    import torch
    from oracle import oracle as om
    cfg = _extreme_cfg(0.5, 25.0)
    env, orc = hip.HipBatch(cfg), om.Oracle(cfg)
    seeds = [5, 6]
    np.testing.assert_allclose(env.reset(seeds=seeds).cpu().numpy(), orc.reset(seeds=seeds), rtol=0, atol=2e-4)
    a = np.ones((2, 16), np.float32)
    a[:, ::2] = -1.0                                   # "wind" action: targets +-45 deg, rate-limited 1 deg / step
    for i in range(120):
        obs, rew, _, _ = env.step(torch.as_tensor(a, device="cuda"))
        oo, orew, _, _ = orc.step(a)
        np.testing.assert_allclose(obs.cpu().numpy(), oo, rtol=0, atol=2e-4)
        np.testing.assert_allclose(rew.cpu().numpy(), orew, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(env.info("rotor_uvw_agent").cpu().numpy()[..., 0], orc.info("rotor_uvw_agent")[..., 0],
                               rtol=1e-4)
    assert np.abs(env.info("yaw_agent").cpu().numpy()).max() > 40
    env.check()


def test_u16_record_saturation_is_never_silent(hip):
    """Ranges that could saturate the record are refused at wg_create; a wind override that drives a record out of
    range at run time latches WG_ERR_RANGE, reported by check()."""
    import torch
    with pytest.raises(ValueError, match="16-bit emission record"):
        hip.HipBatch(_extreme_cfg(0.6, 10.0))
    with pytest.raises(ValueError, match="16-bit emission record"):
        hip.HipBatch(_extreme_cfg(0.1, 41.0))
    env = hip.HipBatch(_extreme_cfg(0.1, 10.0))
    env.set_wind(ti=0.7)                                # TI override beyond what k's unorm16 covers
    env.reset(seeds=[1, 2])
    env.step(torch.zeros((2, 16), device="cuda"))
    with pytest.raises(hip.WindGymHipError, match="16-bit range"):
        env.check()
