"""Oracle spot checks AT BENCH BATCH SIZE.  The step-for-step oracle tests of test_gpu_parity.py run a handful
of envs; the soaks at BASELINE's batch sizes check properties only.  Here the FULL batch of each GPU config runs on the HIP
path — the kernels, block orders and occupancies the bench line times (cfg2: one wave per env with the glue fused, 4096 envs,
and the three-waves-per-env instantiation of 512 / 1024 envs;
cfg3: 256-thread large-farm variant, 512 envs; cfg4: per-agent buffer, 2048 envs; cfg5: k_flow_envb, the one-launch frozen-box
kernel, 1024 envs on a small box) — and the CPU oracle replays 16 of its envs, spread over the batch, on the same global seeds and actions: every step
for 300 steps, through at least one rollover of each sampled env where the episode length allows, with the bars of DESIGN.md §6.
(test_results_do_not_depend_on_batch_composition shows an env does not depend on its neighbours: the subset is representative.)"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

OBS_ATOL = 2e-4
TURB_OBS_ATOL = 5e-4
N_SAMPLE = 16
STEPS = 300


@pytest.fixture(scope="module")
def hip():
    import torch
    from windgym_amd import binding
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    binding.load_library()
    return binding


def _cfgs(workload, B, n_passthrough):
    import bench
    full, sub = bench.make_cfg(B, workload=workload), bench.make_cfg(N_SAMPLE, workload=workload)
    full.n_passthrough = sub.n_passthrough = n_passthrough
    return full, sub


def _run(hip, oracle_lib, workload, B, n_passthrough, variant, turbulent=False, multi=False, min_rollovers=N_SAMPLE, steps=STEPS):
    import torch
    cfg, sub = _cfgs(workload, B, n_passthrough)
    env, orc = hip.HipBatch(cfg), oracle_lib.Oracle(sub)
    assert env.flow_variant() == variant, env.flow_variant()
    if turbulent:
        from windgym_amd.mann import generate_mann_box
        box, spacing = generate_mann_box((256, 64, 32), (3.0, 3.0, 3.0), seed=1234), (3.0, 3.0, 3.0)
        env.set_turbulence_box(box, spacing), orc.set_turbulence_box(box, spacing)
    mbuf = env.fuse_obs_multi() if multi else None
    idx = np.linspace(0, B - 1, N_SAMPLE).round().astype(int)          # first, last and 14 envs in between
    seeds = 1234 + np.arange(B)                                        # bench.py's seeding
    atol = TURB_OBS_ATOL if turbulent else OBS_ATOL
    obs0 = env.reset(seeds=seeds).cpu().numpy()
    np.testing.assert_allclose(obs0[idx], orc.reset(seeds=seeds[idx]), rtol=0, atol=atol)
    for k in ("ws_global", "wd_global", "ti_global"):
        np.testing.assert_allclose(env.info(k).cpu().numpy()[idx], orc.info(k), rtol=1e-7)
    np.testing.assert_array_equal(env.info("time_max").cpu().numpy()[idx], orc.info("time_max").astype(int))
    g = torch.Generator(device="cpu").manual_seed(0)
    n_tr = np.zeros(N_SAMPLE, dtype=int)
    idx_t = torch.as_tensor(idx, device="cuda")          # (the sampled rows are picked on the device: 16 rows cross PCIe, not B)
    for step in range(steps):
        a = torch.rand((B, cfg.n_turb), generator=g) * 2 - 1
        obs, rew, tr, fin = env.step(a.cuda())
        o_obs, o_rew, o_tr, o_fin = orc.step(a.numpy()[idx])
        np.testing.assert_array_equal(tr[idx_t].cpu().numpy().astype(bool), o_tr, err_msg=f"step {step}")
        np.testing.assert_allclose(obs[idx_t].cpu().numpy(), o_obs, rtol=0, atol=atol, err_msg=f"obs step {step}")
        np.testing.assert_allclose(fin[idx_t].cpu().numpy(), o_fin, rtol=0, atol=atol, err_msg=f"final obs step {step}")
        np.testing.assert_allclose(rew[idx_t].cpu().numpy(), o_rew, rtol=1e-3 if turbulent else 1e-4, atol=1e-3 if turbulent else OBS_ATOL,
                                   err_msg=f"reward step {step}")
        n_tr += o_tr.astype(int)
        if step % 25 == 0 or step == steps - 1:
            np.testing.assert_allclose(env.info("yaw_agent").cpu().numpy()[idx], orc.info("yaw_agent"), atol=1e-4)
            np.testing.assert_allclose(env.info("rotor_uvw_agent").cpu().numpy()[idx], orc.info("rotor_uvw_agent"),
                                       rtol=2e-3 if turbulent else 1e-4, atol=2e-3 if turbulent else 1e-4, err_msg=f"rotor wind step {step}")
            np.testing.assert_allclose(env.info("power_turb_agent").cpu().numpy()[idx], orc.info("power_turb_agent"),
                                       rtol=5e-3 if turbulent else 4e-4, atol=2000.0 if turbulent else 20.0, err_msg=f"power step {step}")
            # (power follows the cube of the rotor wind speed below rated: 3 x the 1e-4 of the line above, + the table's kinks)
            if multi:
                np.testing.assert_allclose(mbuf.cpu().numpy()[idx], orc.obs_multi(), rtol=0, atol=atol, err_msg=f"per-agent obs step {step}")
    env.check()
    assert (n_tr >= 1).sum() >= min_rollovers, n_tr
    env.close()


def test_cfg2_4096_envs_one_wave_per_env_fused_glue(hip, oracle_lib):
    """the headline batch: 4x4 farm x 4096 envs x 2 farms, k_flow_env with the glue as its tail (one launch per step)"""
    _run(hip, oracle_lib, "cfg2", 4096, 1.0, (64, True, 2))


def test_cfg2_4096_envs_on_the_bench_schedule(hip, oracle_lib):
    """the headline batch on the schedule the bench line times — n_passthrough = 5 (episodes of 426-1121
    steps, background development spread over them: a fifth of the share per launch the 1.0 runs above give it) — 1250 steps, so
    that every sampled env rolls over under that schedule; the oracle replays its 16 envs at the same length."""
    _run(hip, oracle_lib, "cfg2", 4096, 5.0, (64, True, 2), steps=1250)


@pytest.mark.parametrize("B", [512, 1024])
def test_cfg2_small_batches_with_the_pass_wave(hip, oracle_lib, B):
    """the small-batch instantiation (three waves per env: the running episode's advection pass on a wave of its own, its stores
    behind an LDS flag) at the two batch sizes VERDICT r5 item 6 names, against the oracle through rollovers"""
    _run(hip, oracle_lib, "cfg2", B, 1.0, (64, True, 2))


def test_cfg3_512_envs_large_farm_variant(hip, oracle_lib):
    """Horns Rev 1 x 512 envs (the per-GPU share of BASELINE's 4096 over 8): episodes of 168-360 steps at n_passthrough 0.5"""
    _run(hip, oracle_lib, "cfg3", 512, 0.5, (256, True, 0), min_rollovers=8)


def test_cfg4_2048_envs_per_agent_buffer(hip, oracle_lib):
    """3x3 PettingZoo farm x 2048 envs, per-agent observations written by the fused step into the registered buffer"""
    _run(hip, oracle_lib, "cfg4", 2048, 1.0, (64, True, 2), multi=True)


def test_cfg5_1024_envs_frozen_box(hip, oracle_lib):
    """frozen Mann box + meandering + wake-added turbulence, 16 turbines x 1024 envs, on a small box the oracle shares"""
    _run(hip, oracle_lib, "cfg5", 1024, 1.0, (64, True, 2), turbulent=True)
