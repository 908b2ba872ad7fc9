#!/usr/bin/env python3
"""Extract the only DYNAMIKS-produced numbers that exist in the reference tree into tests/golden/dynamiks_anchors.json.

Run only in the dev container (reads /root/reference):

    python tests/golden/make_dynamiks_anchors.py

Two artefacts of real DYNAMIKS-backed runs of examples/EnvConfigs/Env1.yaml (2 x 2 V80, 8 D pitch) ship with the
reference; neither is asserted by any of its tests:
  * examples/"Example 1 Make environment.ipynb", output of cell 4: the info dict 20 random-action steps into an episode
    (global wind, per-turbine rotor wind speed / direction / power / yaw of the agent and the baseline farm, the 25-step
    sensor means, the flow-frame turbine positions);
  * examples/PPO_2975000.zip -> data["_last_obs"]: the last observation (16 envs x 8) of a PPO run — per turbine the
    25-step mean rotor wind speed and the 10-step mean yaw, scaled to [-1, 1] with the Env1.yaml ranges
    (ws: 2..25 m/s, yaw: -45..45 deg; Wind_Farm_Env.py:440-447).
Only numbers are written.  The wind conditions of the 16 PPO envs are not recorded anywhere (they are nuisance
parameters for tests/test_dynamiks_anchors.py).
"""
import base64
import json
import os
import pickle
import re
import zipfile

import numpy as np

REF = "/root/reference/examples"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dynamiks_anchors.json")


def notebook_info():
    nb = json.load(open(os.path.join(REF, "Example 1 Make environment.ipynb")))
    txt = None
    for c in nb["cells"]:
        if c["cell_type"] == "code" and "".join(c["source"]).strip().endswith("info"):
            for o in c.get("outputs", []):
                if "data" in o and "text/plain" in o["data"]:
                    txt = "".join(o["data"]["text/plain"])
    assert txt is not None
    env = {"array": lambda x, dtype=None: list(map(float, np.asarray(x).ravel())), "float32": "float32", "np": np}
    info = eval(txt, env)                      # a printed dict of floats / arrays
    return {k: (float(v) if np.isscalar(v) else v) for k, v in info.items()}


def ppo_last_obs():
    with zipfile.ZipFile(os.path.join(REF, "PPO_2975000.zip")) as z:
        data = json.loads(z.read("data"))
        sysinfo = z.read("system_info.txt").decode()
    obs = pickle.loads(base64.b64decode(data["_last_obs"][":serialized:"]))
    obs = np.asarray(obs, dtype=np.float64)
    return dict(obs=obs.tolist(), n_envs=int(data["n_envs"]), num_timesteps=int(data["num_timesteps"]),
                ws=((obs[:, 0::2] + 1) / 2 * 23 + 2).tolist(), yaw=((obs[:, 1::2] + 1) / 2 * 90 - 45).tolist(),
                sb3=re.sub(r"\s+", " ", sysinfo).strip())


def main():
    out = dict(source="DTUWindEnergy/WindGym examples (DYNAMIKS @77f4f87 behind WindFarmEnv, Env1.yaml)",
               layout=dict(nx=2, ny=2, pitch_D=8.0, D=80.0, order="turbine 1 is 8 D behind 0, turbine 3 behind 2 (wd 270)"),
               notebook=notebook_info(), ppo=ppo_last_obs())
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    nbk = out["notebook"]
    print("notebook: ws", nbk["Wind speed Global"], "turbines", nbk["Wind speed at turbines"])
    print("ppo _last_obs:", np.asarray(out["ppo"]["obs"]).shape)


if __name__ == "__main__":
    main()
