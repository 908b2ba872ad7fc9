# usage (GPU box): WL=cfg2 bash tools/timeline2.sh  -> per-phase cycles of k_flow's live workgroups (steady compact variant, pair phase first)
cd $GRAFT_REPO_ROOT
cp windgym_amd/libwindgym_hip.so /tmp/lib_keep.so
WG_HIPCC_FLAGS="-DWG_TIMELINE $XF" python windgym_amd/build.py > /dev/null 2>&1
WG_TIMELINE_OUT=gpurun_out/timeline.bin python bench.py --workload ${WL:-cfg2} --steps 60 --warmup 10 --reps 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'kflow ms', d['roofline']['kernel_ms'])"
cp /tmp/lib_keep.so windgym_amd/libwindgym_hip.so
python - <<'PY'
import numpy as np
raw = np.fromfile('gpurun_out/timeline.bin', dtype=np.int64).reshape(-1, 16)
ok = (raw[:, 8] > raw[:, 0]) & (raw[:, 0] > 0) & (raw[:, 9] > raw[:, 2]) & (raw[:, 11] > raw[:, 1])
a = raw[ok]
print('blocks with one full step:', len(a), 'of', len(raw))
seq = [(0, 1, 'prologue: kernarg + state loads + LDS set-up'), (1, 4, 'step set-up (controller, clocks, parameters)'), (4, 5, 'candidate pass'), (5, 11, 'brackets + LDS-DMA gathers issued'), (11, 2, 'records'),
       (2, 10, 'quad list'), (10, 9, 'deficits from the landed gathers + sums'),
       (9, 3, 'advection pass (loads, compute, stores issued)'), (3, 6, 'clock update'), (6, 7, 'tail (power, measurement, ring push)'),
       (7, 8, 'epilogue (state stores, accounting)')]
tot = a[:, 8] - a[:, 0]
print('total: mean %.0f median %.0f p90 %.0f' % (tot.mean(), np.median(tot), np.percentile(tot, 90)))
for i, j, n in seq:
    d = a[:, j] - a[:, i]
    print(f'{n:60s} mean {d.mean():8.0f}  median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}')
PY
