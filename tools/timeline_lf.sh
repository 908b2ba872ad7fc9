# usage (GPU box): bash tools/timeline_lf.sh <variant lib name built with -DWG_TIMELINE> ...  -> mean shader-clock offset of every phase
# stamp of the large-farm kernel's workgroups (cfg3, last launch, workgroups that ran one live flow step), in time order
cd $GRAFT_REPO_ROOT
for n in "$@"; do
WG_DEBUG_HOOKS=1 WG_LIB=$PWD/windgym_amd/variants/lib_$n.so WG_TIMELINE_OUT=gpurun_out/timeline_$n.bin python bench.py --workload cfg3 --no-cpu 2>/dev/null | python tools/benchline.py $n | cut -c1-60
python - $n <<'PY'
import sys, numpy as np
raw = np.fromfile('gpurun_out/timeline_%s.bin' % sys.argv[1], dtype=np.int64).reshape(-1, 16)
ok = (raw[:, 8] > raw[:, 0]) & (raw[:, 0] > 0)
a = raw[ok]
tot = a[:, 8] - a[:, 0]
thr = 0.5 * (np.percentile(tot, 10) + np.percentile(tot, 90))
for nm, m in (("long (moving chains)", tot >= thr), ("short (resting chains)", tot < thr)):
    b = a[m]
    if len(b) == 0: continue
    rel = b - b[:, :1]
    mean = rel.mean(axis=0)
    used = [k for k in range(16) if (b[:, k] > b[:, 0]).mean() > 0.9 or k == 0]
    order = sorted(used, key=lambda k: mean[k])
    print(f"  {nm}: {len(b)} workgroups, total {tot[m].mean():.0f} ticks")
    prev = 0.0
    for k in order:
        print(f"    stamp {k:2d} at {mean[k]:9.0f}  (+{mean[k]-prev:8.0f})")
        prev = mean[k]
PY
done
