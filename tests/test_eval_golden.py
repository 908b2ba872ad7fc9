"""Row f3: the dataset ``eval_sweep`` assembles == the dataset the REFERENCE's ``eval_single_fast`` assembles
(WindGym/AgentEval.py:39-477), replayed from golden vectors recorded by importing the reference itself
(tests/golden/make_eval_golden.py: scripted flow double + scripted model): variables and their order, dims order,
coords, the initial snapshot at time[0] (reward 0), ``pct_inc``.  CPU: the host-side assembly over an oracle-backed
stand-in for the HIP batch; GPU: the real batch in replay mode."""
import glob
import json
import os

import numpy as np
import pytest
import torch

from windgym_amd.turbine import V80

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[12:-4] for p in glob.glob(os.path.join(HERE, "eval_single_*.npz")))


class ScriptedModel:
    def __init__(self, actions):
        self.actions, self.k = actions, 0

    def predict(self, obs, deterministic=False):
        a = self.actions[self.k]
        self.k += 1
        return a, None


class OracleBatch:
    """binding.HipBatch's methods over the oracle (CPU tensors): what evaluate.eval_sweep calls."""

    def __init__(self, cfg, device=None):
        from oracle import oracle as om
        self.torch, self.device = torch, torch.device("cpu")
        self.cfg = cfg
        self.orc = om.Oracle(cfg)
        self.B, self.N, self.obs_dim = cfg.n_envs, cfg.n_turb, self.orc.obs_dim

    def set_wind(self, ws=None, wd=None, ti=None):
        c = self.cfg                      # the oracle has no override table: the YAML ranges must already pin the wind
        assert np.all(np.asarray(ws) == c.ws_min) and c.ws_min == c.ws_max
        assert np.all(np.asarray(wd) == c.wd_min) and c.wd_min == c.wd_max
        assert np.all(np.asarray(ti) == c.TI_min) and c.TI_min == c.TI_max

    def set_flow_script(self, uvw, power):
        self.orc.set_flow_script(uvw, power)

    def reset(self, seeds=None, mask=None):
        return torch.as_tensor(self.orc.reset(seeds=seeds, mask=mask), dtype=torch.float32)

    def step(self, actions):
        obs, rew, tr, fin = self.orc.step(actions.numpy())
        return (torch.as_tensor(obs, dtype=torch.float32), torch.as_tensor(rew, dtype=torch.float32), torch.as_tensor(tr),
                torch.as_tensor(fin, dtype=torch.float32))

    def info(self, name):
        return torch.as_tensor(self.orc.info(name))

    def check(self):
        pass

    def close(self):
        pass


def _run(name, monkeypatch=None):
    from windgym_amd import envs, evaluate
    g = np.load(os.path.join(HERE, f"eval_single_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(g["meta"]))
    if monkeypatch is not None:
        monkeypatch.setattr(envs, "HipBatch", OracleBatch)
    cfg = meta["cfg"]
    cfg["wind"] = dict(ws_min=meta["ws"], ws_max=meta["ws"], wd_min=meta["wd"], wd_max=meta["wd"], TI_min=meta["ti"],
                       TI_max=meta["ti"])                        # FarmEval.set_wind_vals (FarmEval.py:63-78)
    T = max(g["script0_uvw"].shape[0], g["script1_uvw"].shape[0])
    pad = lambda a: np.concatenate([a, np.repeat(a[-1:], T - a.shape[0], axis=0)]) if a.shape[0] < T else a
    uvw = np.stack([pad(g["script0_uvw"]), pad(g["script1_uvw"])])[:, :, None]
    pw = np.stack([pad(g["script0_power"]), pad(g["script1_power"])])[:, :, None]
    ds = evaluate.eval_sweep(V80(), None, ScriptedModel(g["actions"]), winddirs=[meta["wd"]], windspeeds=[meta["ws"]],
                             turbintensities=[meta["ti"]], t_sim=meta["t_sim"], turbtype="None", turbbox="Default",
                             model_step=7, Baseline_comp=meta["two_farms"], yaw_init="Zeros", yaml_dict=cfg,
                             n_rotor_pts=4, n_particles=32, flow_script=(uvw, pw), seed=meta["seed"])
    return g, meta, ds


def _compare(g, meta, ds):
    assert isinstance(ds, dict), "xarray is not installed here: the plain-dict form is expected"
    assert list(ds["data"].keys()) == meta["var_order"]                       # same variables, same order
    for k in meta["var_order"]:
        want = g["var__" + k]
        assert list(ds["dims"][k]) == meta["dims"][k], k                       # same dims, same order
        assert ds["data"][k].shape == want.shape, (k, ds["data"][k].shape, want.shape)
        tol = dict(rtol=2e-4, atol=25.0) if k.startswith("power") else dict(rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(ds["data"][k], want, err_msg=k, **tol)
    assert ds["data"]["reward"][0].item() == 0.0                               # no reward at the initial snapshot (:138)
    assert set(ds["coords"].keys()) == set(meta["coord_order"])
    for k in ("ws", "wd", "TI", "turb", "model_step"):
        np.testing.assert_array_equal(np.asarray(ds["coords"][k], dtype=float), g["coord__" + k].astype(float))
    assert list(ds["coords"]["turbbox"]) == [str(x) for x in g["coord__turbbox"]]
    # fs.time: one dt_env per step from the initial snapshot on (the absolute origin includes the reference's
    # fs.run(t_developed), which the replay double only adds up)
    np.testing.assert_allclose(np.diff(ds["coords"]["time"]), np.diff(g["coord__time"]))


@pytest.mark.parametrize("name", CASES)
def test_eval_sweep_assembles_the_reference_dataset_cpu(name, monkeypatch, oracle_lib):
    assert CASES, "golden files missing"
    _compare(*_run(name, monkeypatch))


@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_eval_sweep_assembles_the_reference_dataset_hip(name):
    _compare(*_run(name))


@pytest.mark.gpu
def test_fixed_box_per_env_matches_oracle_and_consumes_no_draw(oracle_lib):
    """wg_set_box_ids = FarmEval.update_tf per env: env e reads box ids[e] of the pool; like np_random.choice over a
    list of ONE file (FarmEval.py:86-90, Wind_Farm_Env.py:614) the fixed choice consumes no random number, so the
    sampled wind of the episode equals that of a run with a single box."""
    from windgym_amd import binding, presets
    from windgym_amd.config import EnvConfig
    from windgym_amd.mann import generate_mann_box
    spacing = (3.0, 3.0, 3.0)
    boxes = [generate_mann_box((128, 64, 32), spacing, seed=s) for s in (1, 2, 3)]
    B = 4
    cfg = EnvConfig(turbine=V80(), yaml_dict=presets.env1_config(), turbtype="MannLoad", n_envs=B, autoreset=False,
                    n_rotor_pts=4)
    ids = np.array([2, 0, 1, 2], dtype=np.int32)
    env, orc = binding.HipBatch(cfg), oracle_lib.Oracle(cfg)
    for e in (env, orc):
        e.set_turbulence_boxes(boxes, spacing)
        e.set_box_ids(ids)
    seeds = 50 + np.arange(B)
    np.testing.assert_allclose(env.reset(seeds=seeds).cpu().numpy(), orc.reset(seeds=seeds), rtol=0, atol=3e-4)
    np.testing.assert_array_equal(env.info("box_id").cpu().numpy(), ids)
    np.testing.assert_array_equal(orc.info("box_id").astype(int), ids)
    # the wind of a single-box run with the same seeds: no draw was consumed by the fixed choice
    one = binding.HipBatch(EnvConfig(turbine=V80(), yaml_dict=presets.env1_config(), turbtype="MannFixed", n_envs=B,
                                     autoreset=False, n_rotor_pts=4))
    one.set_turbulence_box(boxes[0], spacing)
    one.reset(seeds=seeds)
    np.testing.assert_array_equal(env.info("wind_f64").cpu().numpy(), one.info("wind_f64").cpu().numpy())
    a = np.zeros((B, cfg.n_turb), np.float32)
    for _ in range(30):
        obs, *_ = env.step(torch.as_tensor(a, device="cuda"))
        oo, *_ = orc.step(a)
        np.testing.assert_allclose(obs.cpu().numpy(), oo, rtol=0, atol=3e-4)


@pytest.mark.gpu
def test_eval_sweep_over_two_turbulence_boxes():
    """eval_multiple's turbbox loop as one batched rollout: dims (..., turbbox = 2, model_step); the slice of box k equals
    a sweep run with box k alone."""
    from windgym_amd import presets
    from windgym_amd.agents import ConstantAgent
    from windgym_amd.evaluate import eval_sweep
    from windgym_amd.mann import generate_mann_box
    spacing = (3.0, 3.0, 3.0)
    boxes = [(generate_mann_box((128, 64, 32), spacing, seed=s), spacing) for s in (4, 5)]
    kw = dict(winddirs=[265.0, 275.0], windspeeds=[9.0], turbintensities=[0.06], t_sim=25, yaml_dict=presets.env1_config(),
              n_rotor_pts=4, Baseline_comp=True)
    model = ConstantAgent(np.zeros(4))
    both = eval_sweep(V80(), None, model, turbboxes=boxes, **kw)
    assert both["data"]["powerF_a"].shape == (25, 1, 2, 1, 2, 1) and both["data"]["yaw_a"].shape == (25, 4, 1, 2, 1, 2, 1)
    assert list(both["coords"]["turbbox"]) == ["box0", "box1"]
    for k in range(2):
        single = eval_sweep(V80(), None, model, turbboxes=[boxes[k]], **kw)
        for v in ("powerF_a", "ws_a", "powerF_b", "pct_inc"):
            np.testing.assert_allclose(both["data"][v][..., k:k + 1, :], single["data"][v], rtol=1e-5, atol=1e-5, err_msg=v)
    assert np.abs(both["data"]["ws_a"][..., 0, :] - both["data"]["ws_a"][..., 1, :]).max() > 0.05
