"""Kernel-variant A/Bs on identical seeds and actions (these lived in tools/env_vs_gl.py, which the driver
never runs).  Every pair runs the SAME handle configuration through two kernel variants selected with the debug hooks:

  fused   step() as ONE launch (k_flow_env with the glue as its tail)  vs  flow launch + k_glue_lean      -> bit-identical
  wpe     two waves per env (one per context)                          vs  one wave per env               -> bit-identical
  split   a third wave per env runs the advection pass of the          vs  the same two-wave kernel       -> bit-identical,
          running episode's context and fetches the glue's step-              without it                        state blobs too
          independent inputs ahead (default at 513 .. 1024 envs for
          farms with long chains); split_both: a fourth wave does the
          background context's pass (default at <= 512 envs)
  gl      k_flow_env (lane = farm slot x turbine)                      vs  k_flow GL (one farm slot per workgroup)
          same state layout, arithmetic and summation orders; the compiler contracts a few products differently, so the
          flow values agree to the last bits (held to 1e-5 relative over hundreds of steps), decisions (truncation) exactly

at B = 64 and B = 389 (not a multiple of anything) through several episode rollovers, + one 5000-step run of the fused pair
(the glue's header / deque loads are plain global loads now, wg_glue_lean.h: what used to rest on a register pin)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS = ["yaw_agent", "yaw_base", "rotor_uvw_agent", "rotor_uvw_base", "power_turb_agent", "power_turb_base"]


@pytest.fixture(scope="module")
def hip():
    import torch
    from windgym_amd import binding
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    binding.load_library()
    return binding


def _make(hip, d, B, hooks, multi=False, **kw):
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    os.environ.update(hooks)
    try:
        cfg = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None", n_envs=B, autoreset=True, n_rotor_pts=16, **kw)
        env = hip.HipBatch(cfg)
    finally:
        for k in hooks:
            del os.environ[k]
    if multi:
        env.fuse_obs_multi()
    return cfg, env


def _pair(mode):
    """hooks of the (a, b) handles and whether the pair must agree bit for bit"""
    if mode == "fused":
        return {"WG_FLOW_ENV": "1", "WG_STEP_FUSED": "1"}, {"WG_FLOW_ENV": "1", "WG_STEP_FUSED": "0"}, True
    if mode == "wpe":
        return {"WG_FLOW_ENV": "1", "WG_ENV_WPE": "2"}, {"WG_FLOW_ENV": "1", "WG_ENV_WPE": "1"}, True
    if mode in ("split", "split_both"):      # (a pass wave for the running episode's context / for both contexts)
        return ({"WG_FLOW_ENV": "1", "WG_ENV_WPE": "2", "WG_ENV_SPLIT": "1" if mode == "split" else "2"},
                {"WG_FLOW_ENV": "1", "WG_ENV_WPE": "2", "WG_ENV_SPLIT": "0"}, True)
    return {"WG_FLOW_ENV": "1"}, {"WG_FLOW_ENV": "0"}, False


def _cases():
    from windgym_amd import presets
    return {
        "cfg2_4x4": (presets.bench_cfg2_config(), dict(n_passthrough=1, n_particles=128), False),
        "cfg4_3x3_per_agent_buffer": (presets.multi_3x3_config(), dict(n_passthrough=0.5, n_particles=96, extra_timestep_inc=True), True),
        "two_turb_noise_K": (presets.two_turb_config(), dict(n_passthrough=1), False),
        "cfg2_one_farm": (presets._upd(presets.bench_cfg2_config(), power_def=dict(Power_reward="Power_avg")),
                          dict(n_passthrough=1, n_particles=128), False),
    }


def _ab(hip, mode, case, B, steps):
    import torch
    d, kw, multi = _cases()[case]
    ha, hb, exact = _pair(mode)
    cfg, a_env = _make(hip, d, B, ha, multi=multi, **kw)
    _, b_env = _make(hip, d, B, hb, multi=multi, **kw)
    assert a_env.flow_variant()[2] == 2 and b_env.flow_variant()[2] == (0 if mode == "gl" else 2)

    def same(x, y, what, rtol=1e-5, atol=1e-5):
        if exact:
            assert torch.equal(x, y), what
        else:
            fx, fy = x.float(), y.float()
            assert ((fx - fy).abs() <= atol + rtol * fy.abs()).all(), (what, float((fx - fy).abs().max()))

    def same_fields(tag):
        for f in FIELDS:
            try:
                x, y = a_env.info(f), b_env.info(f)
            except Exception:  # noqa: BLE001  (one-farm configurations have no baseline fields)
                continue
            same(x, y, f"{tag}: {f}", atol=20.0 if f.startswith("power") else 1e-5)
    seeds = 900 + np.arange(B)
    same(a_env.reset(seeds=seeds), b_env.reset(seeds=seeds), "reset obs")
    same_fields("after reset")
    g = torch.Generator(device="cpu").manual_seed(5)
    acts = (torch.rand((64, B, cfg.n_turb), generator=g) * 2 - 1).cuda()
    n_tr = 0
    for s in range(steps):
        ra, rb = a_env.step(acts[s % 64]), b_env.step(acts[s % 64])
        assert torch.equal(ra[2], rb[2]), f"truncation flags differ at step {s}"
        same(ra[0], rb[0], f"obs step {s}")
        same(ra[1], rb[1], f"reward step {s}", atol=2e-5)
        same(ra[3], rb[3], f"final obs step {s}")
        if multi:
            same(a_env._multi_buf, b_env._multi_buf, f"per-agent buffer step {s}")
        n_tr += int(ra[2].sum())
        if s % 50 == 49:
            same_fields(f"step {s}")
    a_env.check(); b_env.check()
    if mode in ("fused", "split", "split_both"):
        # particles, rings, headers, window sums: the whole state.  (Not for "wpe": one wave per env defers a retired context's
        # episode set-up behind its step, so the BACKGROUND context's development runs a launch behind the two-wave schedule —
        # every output is equal, the not-yet-live episode's intermediate state is not.)
        sa, sb = a_env.get_state(), b_env.get_state()
        assert np.array_equal(np.frombuffer(sa, np.uint8), np.frombuffer(sb, np.uint8)), "state blobs differ"
    a_env.close(); b_env.close()
    return n_tr


# every case at B = 64; the two bench shapes also at the large odd batch
PAIRS = [(m, c, 64) for m in ("fused", "wpe", "gl") for c in ("cfg2_4x4", "cfg4_3x3_per_agent_buffer", "two_turb_noise_K", "cfg2_one_farm")]
PAIRS += [(m, c, 389) for m in ("fused", "wpe", "gl") for c in ("cfg2_4x4", "cfg4_3x3_per_agent_buffer")]
PAIRS += [(m, c, B) for m in ("split", "split_both") for c in ("cfg2_4x4", "cfg4_3x3_per_agent_buffer", "two_turb_noise_K", "cfg2_one_farm")
          for B in (64, 389)]


@pytest.mark.parametrize("mode,case,B", PAIRS)
def test_variant_pairs_agree(hip, mode, case, B):
    n_tr = _ab(hip, mode, case, B, 400)
    assert n_tr >= B                                   # every env rolled over at least once on average


def test_fused_step_equals_two_launches_over_5000_steps(hip):
    """the fused glue's loads are correct by construction now (no scalar-cache reads of words the launch
    writes) — 5000 steps x 64 envs with ~30 rollovers per env, bit for bit against flow + k_glue_lean."""
    n_tr = _ab(hip, "fused", "cfg2_4x4", 64, 5000)
    assert n_tr >= 20 * 64


@pytest.mark.parametrize("wpe", ["4", "2", "1"])
def test_box_env_kernel_agrees_with_the_per_slot_kernel(hip, wpe):
    """k_flow_envb (wg_envb.hip: one launch per step, lanes turbine-major, sample-major deficit phase) against k_flow<64, BOX>
    (one workgroup per farm slot + k_glue_lean) on the SAME state layout, seeds and actions, frozen Mann box with the
    wake-added field on, through rollovers: decisions (truncations, the box-pool draw) exactly, values to float rounding — the
    two kernels sum the rotor points and the wake-added shares in different orders (held to 2e-5 of the scaled observation;
    both are held to the oracle's bars separately, tests/test_gpu_parity.py)."""
    import torch
    from windgym_amd import presets
    from windgym_amd.config import EnvConfig
    from windgym_amd.mann import generate_mann_box
    from windgym_amd.turbine import V80
    d = presets.bench_cfg2_config()
    B = 48
    box, spacing = generate_mann_box((256, 64, 32), (3.0, 3.0, 3.0), seed=1234), (3.0, 3.0, 3.0)

    def make(hooks):
        os.environ.update(hooks)
        try:
            cfg = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="MannGenerate", n_envs=B, autoreset=True, n_rotor_pts=16,
                            n_passthrough=1, n_particles=128)
            env = hip.HipBatch(cfg)
        finally:
            for k in hooks:
                del os.environ[k]
        env.set_turbulence_box(box, spacing)
        return cfg, env
    cfg, a_env = make({"WG_FLOW_ENV": "1", "WG_ENV_WPE": wpe})
    _, b_env = make({"WG_FLOW_ENV": "0"})
    assert a_env.flow_variant() == (64, True, 2) and b_env.flow_variant() == (64, True, 0)
    seeds = 700 + np.arange(B)
    oa, ob = a_env.reset(seeds=seeds), b_env.reset(seeds=seeds)
    assert (oa - ob).abs().max().item() <= 2e-5
    g = torch.Generator(device="cpu").manual_seed(7)
    acts = (torch.rand((64, B, cfg.n_turb), generator=g) * 2 - 1).cuda()
    n_tr = 0
    for s in range(360):
        ra, rb = a_env.step(acts[s % 64]), b_env.step(acts[s % 64])
        assert torch.equal(ra[2], rb[2]), f"truncation flags differ at step {s}"
        assert (ra[0] - rb[0]).abs().max().item() <= 2e-5, s
        assert (ra[3] - rb[3]).abs().max().item() <= 2e-5, s
        assert ((ra[1] - rb[1]).abs() <= 1e-4 + 1e-4 * rb[1].abs()).all(), s
        n_tr += int(ra[2].sum())
        if s % 60 == 59:
            for f in FIELDS:
                x, y = a_env.info(f).float(), b_env.info(f).float()
                tol = 50.0 if f.startswith("power") else 2e-4
                assert ((x - y).abs() <= tol + 1e-4 * y.abs()).all(), (s, f, float((x - y).abs().max()))
    assert n_tr >= B
    a_env.check(); b_env.check()
    a_env.close(); b_env.close()


def test_stencil_records_equal_the_brick_ordered_box_bit_for_bit(hip):
    """k_flow_envb reads a rotor point's 8 box corners from ONE 128-byte stencil record (FlowPtrs::box8 / abox8: 8 x the box in
    HBM) where the pool is small enough, else cell by cell from the brick-ordered box: the same cells with the same weights
    in the same association — every output and the whole state must be BIT-identical (WG_NO_BOX8 keeps the records from being
    built).  A box whose dimensions are not powers of two: the records wrap their indices in double, the bricks with %."""
    import torch
    from windgym_amd import presets
    from windgym_amd.config import EnvConfig
    from windgym_amd.mann import generate_mann_box
    from windgym_amd.turbine import V80
    d = presets.bench_cfg2_config()
    B = 24
    box, spacing = generate_mann_box((240, 72, 40), (3.0, 3.0, 3.0), seed=77), (3.0, 3.0, 3.0)
    envs = []
    for no8 in (False, True):
        cfg = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="MannGenerate", n_envs=B, autoreset=True, n_rotor_pts=16,
                        n_passthrough=1, n_particles=128)
        env = hip.HipBatch(cfg)
        assert env.flow_variant() == (64, True, 2)
        if no8:
            os.environ["WG_NO_BOX8"] = "1"
        try:
            env.set_turbulence_box(box, spacing)
            from windgym_amd.mann import default_added_box
            env.set_added_turbulence_box(*default_added_box())
        finally:
            os.environ.pop("WG_NO_BOX8", None)
        envs.append(env)
    a_env, b_env = envs
    seeds = 70 + np.arange(B)
    assert torch.equal(a_env.reset(seeds=seeds), b_env.reset(seeds=seeds))
    g = torch.Generator(device="cpu").manual_seed(9)
    acts = (torch.rand((32, B, cfg.n_turb), generator=g) * 2 - 1).cuda()
    n_tr = 0
    for s_ in range(260):
        ra, rb = a_env.step(acts[s_ % 32]), b_env.step(acts[s_ % 32])
        for x, y in zip(ra, rb):
            assert torch.equal(x, y), s_
        n_tr += int(ra[2].sum())
    assert n_tr >= B
    assert torch.equal(a_env.info("rotor_uvw_agent"), b_env.info("rotor_uvw_agent"))
    assert a_env.get_state() == b_env.get_state()
    a_env.check(); b_env.check()
    a_env.close(); b_env.close()
