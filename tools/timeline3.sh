# usage (GPU box): WL=cfg3 XF="-D..." bash tools/timeline3.sh  -> per-phase cycles of the multi-wave compact steady variants (pair phase first)
cd $GRAFT_REPO_ROOT
cp windgym_amd/libwindgym_hip.so /tmp/lib_keep.so
WG_HIPCC_FLAGS="-DWG_TIMELINE $XF" python windgym_amd/build.py > /dev/null 2>&1
WG_TIMELINE_OUT=gpurun_out/timeline.bin python bench.py --workload ${WL:-cfg3} --steps 60 --warmup 10 --reps 1 --no-cpu $BARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'kflow ms', d['roofline']['kernel_ms'])"
cp /tmp/lib_keep.so windgym_amd/libwindgym_hip.so
python - <<'PY'
import numpy as np
raw = np.fromfile('gpurun_out/timeline.bin', dtype=np.int64).reshape(-1, 16)
ok = (raw[:, 8] > raw[:, 0]) & (raw[:, 0] > 0) & (raw[:, 9] > raw[:, 2]) & (raw[:, 10] > raw[:, 9])
a = raw[ok]
print('blocks with one full step:', len(a), 'of', len(raw))
seq = [(0, 1, 'prologue'), (1, 2, 'step set-up + records'), (2, 14, 'pair phase: masks (1)'), (14, 15, 'pair phase: popcount + wave scans'), (15, 5, 'pair phase: offsets + candidate list'), (5, 12, 'pair phase: chunk 0 brackets, gathers landed, advanced'), (12, 13, 'pair phase: chunk 0 eval_pair (thread 0)'), (13, 11, 'pair phase: chunk 0 rest of the workgroup + barrier'), (11, 9, 'pair phase: sums of chunk 0 + the other chunks'), (9, 10, 'quad list'),
       (10, 3, 'advection pass'), (3, 4, 'store wait / barrier'), (4, 6, 'clock update'), (6, 7, 'tail'), (7, 8, 'epilogue')]
tot = a[:, 8] - a[:, 0]
print('total: mean %.0f median %.0f p90 %.0f' % (tot.mean(), np.median(tot), np.percentile(tot, 90)))
for i, j, n in seq:
    d = a[:, j] - a[:, i]
    print(f'{n:60s} mean {d.mean():8.0f}  median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}')
PY
