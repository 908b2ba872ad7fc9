"""GPU box: one fixed scenario (seeds, actions) through step(); every output of every step goes to an .npz — run it under two
builds (WG_LIB=... with WG_DEBUG_HOOKS=1) and compare the files bit for bit:
    python tools/dump_run.py out_a.npz [workload] [n_envs] [steps];  python tools/dump_run.py --compare out_a.npz out_b.npz"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = [k for k in a.files if not np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8))]
    print("arrays:", len(a.files), "differing:", bad if bad else "none -> BIT-IDENTICAL")
    for k in bad:
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print("  ", k, "max |diff|", d.max(), "first step that differs", int(np.argmax(d.reshape(d.shape[0], -1).max(1) > 0)))
    sys.exit(1 if bad else 0)
import torch  # noqa: E402
import bench  # noqa: E402
from windgym_amd import binding  # noqa: E402

out = sys.argv[1]
wl = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
B = int(sys.argv[3]) if len(sys.argv) > 3 else 48
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 500
cfg = bench.make_cfg(B, autoreset=True, farms2=True, workload=wl)
env = binding.HipBatch(cfg, device=0)
obs0 = env.reset(seeds=7 + np.arange(B))
gen = torch.Generator(device="cpu").manual_seed(3)
acts = (torch.rand((16, B, cfg.n_turb), generator=gen) * 2 - 1).to("cuda").contiguous()
O, R, T, F = [obs0.cpu().numpy()], [], [], []
for i in range(steps):
    o, r, t, f = env.step(acts[i % 16])
    O.append(o.cpu().numpy()); R.append(r.cpu().numpy()); T.append(t.cpu().numpy()); F.append(f.cpu().numpy())
env.check()
np.savez(out, obs=np.stack(O), rew=np.stack(R), trunc=np.stack(T), fin=np.stack(F),
         yaw=env.info("yaw_agent").cpu().numpy(), pw=env.info("power_turb_agent").cpu().numpy(), pwb=env.info("power_turb_base").cpu().numpy())
print("wrote", out, "variant", env.flow_variant(), "truncations", int(np.stack(T).sum()))
