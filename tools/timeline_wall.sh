# usage (GPU box): WL=cfg3 bash tools/timeline_wall.sh [bench args]  -> wall-clock life of every k_flow workgroup of the last launch,
# grouped by what it did (episode set-up at its head, first observation at its end, live / background step)
cd $GRAFT_REPO_ROOT
cp windgym_amd/libwindgym_hip.so /tmp/lib_keep.so
WG_HIPCC_FLAGS="-DWG_TIMELINE -DWG_TIMELINE_WALL $XF" python windgym_amd/build.py > /dev/null 2>&1
WG_DEBUG_HOOKS=1 WG_TIMELINE_OUT=gpurun_out/timeline_wall.bin python bench.py --workload ${WL:-cfg3} --steps 60 --warmup 10 --reps 1 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'kflow ms', d['roofline']['kernel_ms'])"
cp /tmp/lib_keep.so windgym_amd/libwindgym_hip.so
python - <<'PY'
import numpy as np
raw = np.fromfile('gpurun_out/timeline_wall.bin', dtype=np.int64).reshape(-1, 16)
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
        print('  %-32s n %5d  life us median %.1f p90 %.1f max %.1f ; end max %.1f' % (name, sel.sum(), np.median(life[sel]), np.percentile(life[sel], 90), life[sel].max(), en[sel].max()))
order = np.argsort(-en)[:10]
print('last to end: end us | life | flags(1 set-up, 2 first obs, 4 live) | flow steps | start')
for i in order:
    print('   %.1f | %.1f | %d | %d | %.1f' % (en[i], life[i], fl[i], a[i, 13], st[i]))
PY
