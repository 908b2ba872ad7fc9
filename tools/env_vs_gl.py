"""GPU box: k_flow_env (one wave per env) against k_flow GL (one workgroup per farm slot) on identical seeds and actions —
every output of step() and the flow state must be BIT-identical (same state layout, arithmetic and summation orders).
usage: python tools/env_vs_gl.py [n_envs] [steps]"""
import os
import sys

import numpy as np

os.environ["WG_DEBUG_HOOKS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from windgym_amd import binding as hip, presets  # noqa: E402
from windgym_amd.config import EnvConfig  # noqa: E402
from windgym_amd.turbine import V80  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 24
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 400


MODE = sys.argv[3] if len(sys.argv) > 3 else "gl"      # "gl": env kernel vs GL; "fused": fused step vs flow + glue launches; "wpe"


def make(d, envw, **kw):
    if MODE == "fused":
        os.environ["WG_STEP_FUSED"] = "1" if envw else "0"
        envw = True
    if MODE == "wpe":      # two waves per env (one per context) against one wave per env, both with the fused glue
        os.environ["WG_ENV_WPE"] = "2" if envw else "1"
        envw = True
    os.environ["WG_FLOW_ENV"] = "1" if envw else "0"
    try:
        cfg = EnvConfig(turbine=V80(), yaml_dict=d, turbtype="None", n_envs=B, autoreset=True, n_rotor_pts=16,
                        **{k: v for k, v in kw.items() if k != "multi"})
        env = hip.HipBatch(cfg)
    finally:
        del os.environ["WG_FLOW_ENV"]
        os.environ.pop("WG_STEP_FUSED", None)
        os.environ.pop("WG_ENV_WPE", None)
    assert env.flow_variant()[2] == (2 if envw else 0), env.flow_variant()
    if kw.pop("multi", False):
        env.fuse_obs_multi()
    return cfg, env


FIELDS = ["yaw_agent", "yaw_base", "rotor_uvw_agent", "rotor_uvw_base", "power_turb_agent", "power_turb_base"]
cases = [("cfg2 4x4", presets.bench_cfg2_config(), dict(n_passthrough=1, n_particles=128)),
         ("cfg4 3x3", presets.multi_3x3_config(), dict(n_passthrough=1, n_particles=96)),
         ("cfg4 3x3 per-agent buffer", presets.multi_3x3_config(), dict(n_passthrough=0.5, n_particles=96, multi=True)),
         ("2turb noise K", presets.two_turb_config(), dict(n_passthrough=1)),
         ("env1 2x2 wind", presets.env1_config(), dict(n_passthrough=1)),
         ("cfg2 F=1", presets._upd(presets.bench_cfg2_config(), power_def=dict(Power_reward="Power_avg")), dict(n_passthrough=1, n_particles=128))]
bad = 0
for name, d, kw in cases:
    try:
        cfg, a_env = make(d, True, **kw)
        _, b_env = make(d, False, **kw)
    except Exception as ex:  # noqa: BLE001
        print(name, "SKIP:", ex)
        continue
    seeds = 900 + np.arange(B)
    o_a, o_b = a_env.reset(seeds=seeds).cpu().numpy(), b_env.reset(seeds=seeds).cpu().numpy()
    ok = np.array_equal(o_a, o_b)
    print(name, "reset obs equal:", ok, "N", cfg.n_turb)
    for f in FIELDS:
        try:
            x, y = a_env.info(f).cpu().numpy(), b_env.info(f).cpu().numpy()
        except Exception:  # noqa: BLE001
            continue
        if not np.array_equal(x, y, equal_nan=True):
            dd = np.abs(x.astype(np.float64) - y)
            idx = np.argwhere(dd > 0)
            print("   after reset:", f, "differs at", len(idx), "of", dd.size, "max", dd.max(), "first", idx[:6].tolist(), x[tuple(idx[0])], y[tuple(idx[0])])
    rng = np.random.default_rng(5)
    n_tr = 0
    first_bad = None
    for s in range(STEPS):
        a = torch.as_tensor(rng.uniform(-1, 1, size=(B, cfg.n_turb)).astype(np.float32), device="cuda")
        ra, rb = a_env.step(a), b_env.step(a)
        for i, (x, y) in enumerate(zip(ra, rb)):
            if not np.array_equal(x.cpu().numpy(), y.cpu().numpy(), equal_nan=True):
                if first_bad is None:
                    first_bad = (s, "out%d" % i, float(np.nanmax(np.abs(x.cpu().numpy().astype(np.float64) - y.cpu().numpy()))))
        if getattr(a_env, "_multi_buf", None) is not None and not np.array_equal(a_env._multi_buf.cpu().numpy(), b_env._multi_buf.cpu().numpy()):
            if first_bad is None:
                first_bad = (s, "multi_buf", 0.0)
        n_tr += int(ra[2].sum())
        if s % 50 == 49 or first_bad:
            for f in FIELDS:
                try:
                    x, y = a_env.info(f).cpu().numpy(), b_env.info(f).cpu().numpy()
                except Exception:  # noqa: BLE001
                    continue
                if not np.array_equal(x, y, equal_nan=True) and first_bad is None:
                    first_bad = (s, f, float(np.nanmax(np.abs(x.astype(np.float64) - y))))
        if first_bad:
            break
    try:
        a_env.check(); b_env.check()
    except Exception as ex:  # noqa: BLE001
        print("  check:", ex)
    sa, sb = a_env.get_state(), b_env.get_state()
    same_state = sa == sb if isinstance(sa, bytes) else np.array_equal(np.frombuffer(sa, np.uint8), np.frombuffer(sb, np.uint8))
    print(f"  steps {s + 1} truncations {n_tr} first mismatch {first_bad} state blobs equal {same_state}")
    bad += first_bad is not None
    a_env.close(); b_env.close()
print("RESULT", "FAIL" if bad else "OK")
