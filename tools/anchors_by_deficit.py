"""Both DYNAMIKS anchors (tests/test_dynamiks_anchors.py) for the three deficit options, on the HIP path.
usage (GPU box): python tools/anchors_by_deficit.py   -> one line per deficit model"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import test_dynamiks_anchors as T          # noqa: E402
from windgym_amd import binding            # noqa: E402
from windgym_amd.mann import generate_mann_box   # noqa: E402


class _Hip(binding.HipBatch):
    def step(self, a):
        return super().step(torch.as_tensor(a, device="cuda"))


box = generate_mann_box(T.BOX_SPEC["dims"], T.BOX_SPEC["spacing"], seed=T.BOX_SPEC["seed"])
orig = T._cfg
print("reference: PPO median %.3f p10 %.3f p90 %.3f ; notebook 0.915 / 0.522" % (
    np.median(T.R_REF), np.percentile(T.R_REF, 10), np.percentile(T.R_REF, 90)))
for dm in ("gaussian", "super_gaussian", "ainslie"):
    def cfgf(yaw, K, wind=None, dm=dm):
        c = orig(yaw, K, wind)
        c.deficit = dm
        return c
    T._cfg = cfgf
    model = np.array([T._ratios(_Hip, box, T.YAW_REF[e], 96, 1000 * e) for e in range(16)])
    lo, hi = np.percentile(model, 2.5, axis=1), np.percentile(model, 97.5, axis=1)
    inside = ((T.R_REF >= lo) & (T.R_REF <= hi)).sum()
    r, _ = T._notebook_band(_Hip, box, 512)
    print("%-15s PPO median %.3f p10 %.3f p90 %.3f min %.3f inside-95%%-band %d/32 | notebook median %.3f band [%.3f, %.3f] min %.3f" % (
        dm, np.median(model), np.percentile(model, 10), np.percentile(model, 90), model.min(), inside,
        np.median(r), np.percentile(r, 2.5), np.percentile(r, 97.5), r.min()))
T._cfg = orig
