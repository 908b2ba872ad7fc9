#!/usr/bin/env python3
"""BASELINE.json configs[0] (cfg1): the reference's own CPU-runnable case — 2 turbines (V80, 8 D apart), ws 8 m/s,
wd 270, inflow "None", the 2turb.yaml sensors (O = 200), ONE env on ONE core.  Timed with the oracle's C port
(fp64 and fp32) — the apples-to-apples analogue of one reference env process.  Prints one JSON line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle as om                     # noqa: E402
from windgym_amd import presets                     # noqa: E402
from windgym_amd.config import EnvConfig            # noqa: E402
from windgym_amd.turbine import V80                 # noqa: E402


def run(precision, seconds=5.0):
    cfg = EnvConfig(turbine=V80(), yaml_dict=presets.two_turb_config(), turbtype="None", n_envs=1, autoreset=True,
                    n_passthrough=5, n_rotor_pts=16)
    orc = om.Oracle(cfg, precision)
    orc.set_threads(1)
    orc.reset(seeds=[1234])
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(64, 1, cfg.n_turb)).astype(np.float32)
    for i in range(20):
        orc.step(acts[i % 64])
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds:
        for i in range(50):
            orc.step(acts[(n + i) % 64])
        n += 50
    el = time.perf_counter() - t0
    return n / el, orc.obs_dim


if __name__ == "__main__":
    om.build()
    f64, odim = run("f64")
    f32, _ = run("f32")
    print(json.dumps({"workload": "cfg1: 2 turbines, 2turb.yaml sensors, inflow None, B=1, one core", "obs_dim": odim,
                      "oracle_f64_env_steps_per_s": f64, "oracle_f32_env_steps_per_s": f32, "cores": 1,
                      "note": "includes the ctypes call overhead of one step() per call; the reference's own "
                              "glue-only cost is 0.5-2.7 ms/step (SURVEY.md §6)"}))
