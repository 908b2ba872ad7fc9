cd $GRAFT_REPO_ROOT
WG_HIPCC_FLAGS="-DWG_TIMELINE $XF" python windgym_amd/build.py > /dev/null 2>&1
WG_TIMELINE_OUT=gpurun_out/timeline.bin python bench.py --workload ${WL:-cfg2} --steps 60 --warmup 10 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['roofline']['kernel_ms'])"
python - <<'PY'
import numpy as np
a = np.fromfile('gpurun_out/timeline.bin', dtype=np.int64).reshape(-1, 12)
ok = a[:, 8] > a[:, 0]
a = a[ok & (a[:,0] > 0)]
print('blocks with a full step:', len(a))
d = np.diff(a[:, :9], axis=1)
names = ['prologue(0-1)','records(1-2)','advect(2-3)','store-wait(3-4)','phaseA(4-5)','phaseB(5-6)','tail(6-7)','epilogue(7-8)']
tot = a[:,8]-a[:,0]
print('clock = s_memtime ticks (100 MHz constant clock?) ; total mean', tot.mean(), 'median', np.median(tot))
for n, col in zip(names, d.T): print(f'{n:18s} mean {col.mean():9.1f}  median {np.median(col):9.1f}  p90 {np.percentile(col,90):9.1f}')
start = a[:,0]-a[:,0].min(); end = a[:,8]-a[:,0].min()
print('launch span (first start -> last end):', end.max(), ' starts p50/p90/max', np.percentile(start,50), np.percentile(start,90), start.max())
PY
