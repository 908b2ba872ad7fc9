"""Host-side cost of step(): enqueue time per call (no sync) vs wall time per step, cfg2 at 4096 envs.

    python tools/host_overhead.py [steps]

Prints, for a few (warmup, steps) pairs, the time the host spends inside the step() calls and the wall time up to
the closing synchronize — the gap between the two is GPU work the host did not have to wait for.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch

import bench
from windgym_amd import binding


def main():
    B = 4096
    cfg = bench.make_cfg(B)
    env = binding.HipBatch(cfg, device=0)
    env.reset(seeds=1234 + np.arange(B))
    acts = (torch.rand((16, B, cfg.n_turb)) * 2 - 1).cuda().contiguous()
    for timing in (0, 4, 0):
        for warm, steps in ((5, 20), (5, 20), (20, 200), (0, 1000)):
            for i in range(warm):
                env.step(acts[i % 16])
            env.kernel_timing(timing)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                env.step(acts[i % 16])
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            f, g, n, _, _ = env.kernel_timing(0)
            print(f"timing={timing} warm={warm:3d} steps={steps:4d}: enqueue {1e6 * (t1 - t0) / steps:7.1f} us/step, "
                  f"wall {1e6 * (t2 - t0) / steps:7.1f} us/step, kernels {1e3 * (f + g):6.1f} us (n={n})", flush=True)


if __name__ == "__main__":
    main()
