# usage (GPU box): WL=cfg2 XF="-D..." bash tools/timeline_kflow.sh   -> per-phase latency budget of k_flow's workgroups
cd $GRAFT_REPO_ROOT
WG_HIPCC_FLAGS="-DWG_TIMELINE $XF" python windgym_amd/build.py > /dev/null 2>&1
WG_TIMELINE_OUT=gpurun_out/timeline.bin python bench.py --workload ${WL:-cfg2} --steps 60 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'])"
python - <<'PY'
import numpy as np
raw = np.fromfile('gpurun_out/timeline.bin', dtype=np.int64).reshape(-1, 16)
bid = np.arange(len(raw))
ok = (raw[:, 8] > raw[:, 0]) & (raw[:, 0] > 0)
a = raw[ok]; b = bid[ok]
print('blocks with a full step:', len(a), 'of', len(raw))
d = np.diff(a[:, :9], axis=1)
names = ['prologue(0-1)','records(1-2)','advect(2-3)','store-wait(3-4)','phaseA(4-5)','phaseB(5-6)','tail(6-7)','epilogue(7-8)']
tot = a[:,8]-a[:,0]
print('shader-clock ticks; total mean', tot.mean(), 'median', np.median(tot))
for n, col in zip(names, d.T): print(f'{n:18s} mean {col.mean():9.1f}  median {np.median(col):9.1f}  p90 {np.percentile(col,90):9.1f}')
# per-XCD picture of the LAST launch (workgroup i runs on XCD i % 8; clocks of different XCDs are not comparable;
# stamps of workgroups that were idle in the last launch are stale -> keep the final cluster of start times)
for x in range(8):
    m = (b % 8) == x
    if m.sum() == 0: continue
    s0 = a[m, 0]; e0 = a[m, 8]; tt = tot[m]
    o = np.argsort(s0); s0, e0, tt = s0[o], e0[o], tt[o]
    gaps = np.nonzero(np.diff(s0) > 25000)[0]
    k = gaps[-1] + 1 if len(gaps) else 0
    s1, e1, t1 = s0[k:], e0[k:], tt[k:]
    span = e1.max() - s1.min()
    print(f'XCD {x}: active blocks {len(s1):5d}  span {span:8d} ticks  mean block {t1.mean():7.0f}  resident = sum/span {t1.sum()/span:6.1f}  '
          f'last start at {100*(s1.max()-s1.min())/span:5.1f}% of span')
PY
