"""Long soak run of every GPU config (dozens of episode rollovers per env, background episodes, same-step autoreset):
    python tools/soak.py      # on an MI355X; prints one 'ok' line per config
"""
import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import bench
from windgym_amd import binding
for wl, B, steps in (("cfg2", 4096, 20000), ("cfg4", 2048, 20000), ("cfg3", 256, 12000), ("cfg5", 512, 6000)):
    cfg = bench.make_cfg(B, autoreset=True, farms2=True, workload=wl)
    env = binding.HipBatch(cfg, device=0)
    if wl == "cfg5":
        from windgym_amd.mann import generate_mann_box_torch
        env.set_turbulence_box(generate_mann_box_torch(Nxyz=(1024, 256, 64), device="cuda"), (3.0, 3.0, 3.0))
    env.reset(seeds=1 + np.arange(B))
    acts = (torch.rand((16, B, cfg.n_turb), device="cuda") * 2 - 1).contiguous()
    n_tr = 0
    t0 = time.time()
    for i in range(steps):
        obs, rew, tr, fin = env.step(acts[i % 16])
        if i % 500 == 499:
            env.check()
            n_tr += int(tr.sum().item())
            assert torch.isfinite(obs).all() and torch.isfinite(rew).all()
    env.check()
    ep = env.info("episode").cpu().numpy()
    print(wl, "ok:", steps, "steps,", B, "envs, episodes per env min/mean/max", ep.min(), ep.mean(), ep.max(), "in", round(time.time() - t0, 1), "s")
    env.close()
