# usage (GPU box): WL=cfg3 bash tools/timeline_wall_lib.sh <variant built with -DWG_TIMELINE -DWG_TIMELINE_WALL> ...  -> wall-clock life of
# every k_flow workgroup of the last launch (prebuilt variant libraries: tools/fastbuild.sh windgym_amd/variants/lib_<name>.so -D...)
cd $GRAFT_REPO_ROOT
for n in "$@"; do
WG_DEBUG_HOOKS=1 WG_LIB=$PWD/windgym_amd/variants/lib_$n.so WG_TIMELINE_OUT=gpurun_out/timeline_wall_$n.bin python bench.py --workload ${WL:-cfg3} --reps 1 --no-cpu 2>/dev/null | python tools/benchline.py $n | cut -c1-70
python - $n <<'PY'
import sys, numpy as np
raw = np.fromfile('gpurun_out/timeline_wall_%s.bin' % sys.argv[1], dtype=np.int64).reshape(-1, 16)
a = raw[(raw[:, 0] == 1) & (raw[:, 15] > raw[:, 14])]
print('workgroups that took a step in the last launch:', len(a), 'of', len(raw))
t0 = a[:, 14].min()
st, en = (a[:, 14] - t0) / 100.0, (a[:, 15] - t0) / 100.0
life = en - st
print('starts us: median %.1f p90 %.1f max %.1f ; ends: median %.1f p90 %.1f p99 %.1f max %.1f ; life median %.1f p90 %.1f max %.1f' % (
    np.median(st), np.percentile(st, 90), st.max(), np.median(en), np.percentile(en, 90), np.percentile(en, 99), en.max(), np.median(life), np.percentile(life, 90), life.max()))
fl = a[:, 12]
for name, sel in (('episode set-up at the head', (fl & 1) != 0), ('first observation at the end', (fl & 2) != 0), ('live step', (fl & 4) != 0),
                  ('background step', (fl & 4) == 0), ('two or more flow steps', a[:, 13] >= 2)):
    if sel.any():
        print('  %-32s n %5d  life us median %.1f p90 %.1f max %.1f ; start median %.1f max %.1f ; end max %.1f' % (name, sel.sum(), np.median(life[sel]), np.percentile(life[sel], 90), life[sel].max(), np.median(st[sel]), st[sel].max(), en[sel].max()))
ts = np.linspace(0, en.max(), 13)[1:-1]
print('  alive at', ' '.join('%.0f' % t for t in ts), 'us:', [int(((st <= t) & (en > t)).sum()) for t in ts])
print('  sum of lives %.0f us ; / 1024 slots = %.1f us' % (life.sum(), life.sum() / 1024))
PY
done
