import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench
from windgym_amd import binding
cfg = bench.make_cfg(4096)
env = binding.HipBatch(cfg, device=0)
env.reset(seeds=1234 + np.arange(4096))
acts = (torch.rand((16, 4096, 16)) * 2 - 1).cuda()
for i in range(30): env.step(acts[i % 16])
for timing in (False, True, False, True):
    env.kernel_timing(timing)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(300): env.step(acts[i % 16])
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print('timing', timing, 'ms/step', el / 300 * 1e3, env.kernel_timing(False)[:2])
