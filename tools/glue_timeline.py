"""Per-wave phase timeline of k_glue (library built with -DWG_GLUE_TL: shader-clock stamps of every env's wave are left in
its final_obs row).  usage (GPU box): WG_DEBUG_HOOKS=1 WG_LIB=<lib> python tools/glue_timeline.py [cfg2|cfg3|cfg4]"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from windgym_amd import binding

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = {"cfg2": 4096, "cfg3": 512, "cfg4": 2048, "cfg5": 1024}[wl]
cfg = bench.make_cfg(B, autoreset=True, farms2=True, workload=wl)
env = binding.HipBatch(cfg, device=0)
if wl == "cfg4":
    env.fuse_obs_multi()
env.reset(seeds=1234 + np.arange(B))
gen = torch.Generator().manual_seed(0)
acts = [(torch.rand((B, cfg.n_turb), generator=gen) * 2 - 1).cuda() for _ in range(8)]
for i in range(600 if wl != "cfg3" else 150):
    env.step(acts[i % 8])
names = ["headers + loads + deques", "stage + first observation", "reward, metrics", "truncation block (swap, 2nd observation)",
         "background plan", "write-back"]
rows = {0: [], 1: []}
ends = {0: [], 1: []}
kern = []
for i in range(200):
    _, _, tr, fin = env.step(acts[i % 8])
    torch.cuda.synchronize()
    st = fin[:, :9].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    t0 = st[:, 7].min()                          # device-wide 100 MHz clock: 10 ns units
    d = (st[:, 1:6] - st[:, 0:5]) & 0xFFFFFFFF
    end = (st[:, 8] - t0) & 0xFFFFFFFF
    start = (st[:, 7] - t0) & 0xFFFFFFFF
    trn = st[:, 6] != 0
    kern.append(end.max())
    for k in (0, 1):
        m = trn == bool(k)
        if m.any():
            rows[k].append(np.concatenate([start[m, None], d[m], end[m, None]], axis=1))
for k, label in ((0, "waves that do not truncate"), (1, "truncating waves")):
    a = np.concatenate(rows[k])
    print(f"{label}: {len(a)} samples; start after the first wave's start: mean {a[:, 0].mean() / 100:.2f} us p99 {np.percentile(a[:, 0], 99) / 100:.2f}; end: mean {a[:, -1].mean() / 100:.2f} us p99 {np.percentile(a[:, -1], 99) / 100:.2f} max {a[:, -1].max() / 100:.2f}")
    for j, nm in enumerate(names[:5]):
        print(f"   {nm:45s} mean {a[:, 1 + j].mean():8.0f}  p90 {np.percentile(a[:, 1 + j], 90):8.0f}")
print(f"last wave's end after the first wave's start, per launch: mean {np.mean(kern) / 100:.2f} us")
ends_nt = [r[:, -1].max() for r in rows[0]]
print(f"last NON-truncating wave's end, per launch: mean {np.mean(ends_nt) / 100:.2f} us")
