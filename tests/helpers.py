"""Shared helpers for the parity tests: golden fixtures -> EnvConfig, scripted tables, comparisons."""
import glob
import json
import os

import numpy as np

from windgym_amd.config import EnvConfig
from windgym_amd.turbine import V80

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_cases():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "glue_*.npz")))


def load_golden(name):
    g = np.load(os.path.join(GOLDEN, f"glue_{name}.npz"), allow_pickle=False)
    meta = json.loads(str(g["meta"]))
    return g, meta


def config_from_meta(meta, n_envs=1, autoreset=False, **over):
    kw = dict(meta["kwargs"])
    kw.update(over)
    return EnvConfig(turbine=V80(), yaml_dict=meta["cfg"], turbtype="None", n_envs=n_envs, autoreset=autoreset,
                     extra_timestep_inc=bool(meta["multi"]), n_rotor_pts=4, n_particles=32, **kw)


def script_tables(g, n_envs=1):
    """[F,T,B,N,(3)] tables from the golden script; every env of the batch replays the same rows."""
    # the PettingZoo facade resets once inside its constructor (consuming rows) before the recorded reset
    c0, c1 = int(g["cursor0"][0]), int(g["cursor1"][0])
    s0u, s0p, s1u, s1p = g["script0_uvw"][c0:], g["script0_power"][c0:], g["script1_uvw"][c1:], g["script1_power"][c1:]
    T = max(s0u.shape[0], s1u.shape[0])

    def pad(a):
        if a.shape[0] < T:
            a = np.concatenate([a, np.repeat(a[-1:], T - a.shape[0], axis=0)], axis=0)
        return a

    uvw = np.stack([pad(s0u), pad(s1u)])[:, :, None]      # [2,T,1,N,3]
    pw = np.stack([pad(s0p), pad(s1p)])[:, :, None]       # [2,T,1,N]
    uvw = np.repeat(uvw, n_envs, axis=2)
    pw = np.repeat(pw, n_envs, axis=2)
    return np.ascontiguousarray(uvw), np.ascontiguousarray(pw)
